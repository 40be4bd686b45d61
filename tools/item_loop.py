"""The reference's loop body with main.py:102's per-step loss.cpu().item() at the bench shape: ms per step of the plain loop, of
the loop with the host read (published loss), and where the host's time goes (tools/item_loop.py [steps])."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    args = None
    device = torch.device("cuda", 0)
    cfg, ds, model = bench.build(args, device)
    model = model.to(device)
    from elimrec_amd import ColumnShardEngine, ColumnShardTrainer, FusedAdam, PairwiseSamplerV2
    B = bench.WORKLOAD["batch_size"]
    opt = FusedAdam(model.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    eng = ColumnShardEngine(model)
    tr = ColumnShardTrainer(eng, opt)
    sampler = PairwiseSamplerV2(ds, batch_size=B, device=device, seed=cfg["seed"])
    U_, P_, N_ = sampler.sample_epoch()
    nb = min(steps, U_.numel() // B)
    batches = [(U_[i * B:(i + 1) * B], P_[i * B:(i + 1) * B], N_[i * B:(i + 1) * B]) for i in range(nb)]
    tr.prestage(batches)

    def body(k):
        u, p, n = batches[k % nb]
        loss = model.bpr_loss(u, p, n)
        opt.zero_grad()
        loss.backward(retain_graph=True)
        opt.step()
        return loss
    for k in range(30):
        body(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        body(k)
    torch.cuda.synchronize()
    print("plain loop           %.4f ms/step" % (1e3 * (time.perf_counter() - t0) / steps))
    for k in range(30):
        body(k).cpu().item()
    torch.cuda.synchronize()
    waits, t0 = 0.0, time.perf_counter()
    for k in range(steps):
        loss = body(k)
        t1 = time.perf_counter()
        loss.cpu().item()
        waits += time.perf_counter() - t1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = tr._native_state()
    print("with .cpu().item()   %.4f ms/step  (host waits in the read %.4f ms/step, the rest of the host loop %.4f ms/step)"
          % (1e3 * dt, 1e3 * waits / steps, 1e3 * (dt - waits / steps)))
    print("published steps %d, native steps %d, failed=%r, programs=%d" % (model.plugin.published_steps, st["native_steps"], st["failed"], len(st["programs"])))


if __name__ == "__main__":
    main()
