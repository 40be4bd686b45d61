#!/usr/bin/env python3
"""NeuRec-style driver with the reference's command line (reference main.py:27-175):

    python main.py --recommender=EliMRec --data.input.dataset=<name> --alpha=0.5 --loss=bpr_loss [--key=value ...]

Behaviour kept from the reference's `Net.run`: one pass of the pairwise sampler per epoch, validation every
`test_step` epochs with predict_type TIE, a checkpoint + TE/TIE test pass whenever validation recall improves
(not on epoch 0), early stop after `stop_cnt` epochs without improvement, the same log lines. The per-batch work,
the sampler and the evaluator run on the GPU (elimrec_amd). `--data.input.dataset=synthetic` uses the seeded
Tiktok-shape generator instead of reading files.

`--loss=<method>` names the model method that computes the loss, as in the reference (main.py:98), and the epoch loop is the
reference's: loss method -> zero_grad -> backward(retain_graph=True) -> optimizer step. With `bpr_loss` (EliMRec's own loss)
those four lines run the column-shard engine's fused step (elimrec_amd/plugin.py); any other method of the model (`infonce`,
`fast_loss`, the base class's generic losses) goes through the differentiable table build.
`--resume=<checkpoint>` (not in the reference, which only saves): restores the parameters from a reference-format
checkpoint and, when the side file `<checkpoint>.resume` written next to it exists, the Adam moments, step counts,
epoch counter and best metrics.

Multi-GPU: `--gpus=N` (starts N ranks itself) or a torch.distributed.run launch, one process per GPU. Every rank owns recdim/world columns of the
embedding tables (elimrec_amd/shard.py); an epoch is split over the ranks -- each draws 1/world of the epoch's
triplets in batches of batch_size/world, so the global batch, the number of optimizer steps per epoch and the
learning-rate schedule are those of the single-GPU (and the reference's) configuration; the logged loss is the mean
over ranks.
"""
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")      # before the HIP runtime initialises (elimrec_amd/__init__.py has the measurements)

import torch

from elimrec_amd import (Configurator, Dataset, EliMRec, FusedAdam, Logger, Meter, PairwiseSamplerV2,
                         SyntheticDataset, set_seed)
from elimrec_amd import ColumnShardEngine, ColumnShardTrainer
from elimrec_amd.dist import DataParallelTrainer

EFFECTS = ("TE", "TIE")


def count_parameters(module):
    sizes = [(p.numel(), p.requires_grad) for p in module.parameters()]
    return {"Total": sum(n for n, _ in sizes), "Trainable": sum(n for n, g in sizes if g)}


def open_device(cfg):
    """This process's GPU (LOCAL_RANK under torch.distributed.run) and its rank / world size."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if cfg.no_cuda or not torch.cuda.is_available():
        print("use cpu")
        raise SystemExit("elimrec_amd has no CPU path for training: an MI355X is required (no_cuda must be FALSE)")
    same_gpu = os.environ.get("ELIMREC_SAME_GPU") == "1"   # every rank on device 0 (one-GPU staging of the multi-rank job, as in
    if same_gpu:                                           # bench.py: RCCL refuses duplicate devices, so the group is gloo and
        local_rank = 0                                     # the collectives are staged through the host)
    print("use", "cuda:%d" % local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if same_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    return device, rank, world


def open_dataset(cfg):
    if cfg["data.input.dataset"] != "synthetic":
        return Dataset(cfg)
    users, items, interactions = cfg["synthetic_shape"] if "synthetic_shape" in cfg else (36656, 76085, 720829)
    dims = tuple(cfg["synthetic_dims"]) if "synthetic_dims" in cfg else (128, 128, 128)
    return SyntheticDataset(users, items, interactions, feat_dims=dims, seed=0)


class Net(object):
    def __init__(self, args):
        self.config = cfg = args
        Logger.logger = Logger(name="_".join([cfg.recommender, cfg["data.input.dataset"], cfg.loss, cfg.suffix]),
                               show_in_console=cfg.verbose != 0, is_creat_log_file=cfg.create_log_file, path=cfg.log_path)
        Logger.info(cfg.params_str())
        cfg.device, self.rank, self.world = open_device(cfg)
        self.dataset = open_dataset(cfg)
        if cfg.recommender != "EliMRec":
            raise ValueError("unknown recommender '%s'" % cfg.recommender)
        self.recommender = EliMRec(cfg, self.dataset).to(cfg.device)
        self.cf_mode = cfg["cf_mode"] if "cf_mode" in cfg else True
        Logger.info(count_parameters(self.recommender))
        self.opt = FusedAdam(self.recommender.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
        self.loss_name = str(cfg.loss)
        if not callable(getattr(self.recommender, self.loss_name, None)):
            raise AttributeError("'%s' has no loss method '%s' (--loss)" % (cfg.recommender, self.loss_name))
        self.engine = self.trainer = None
        rec = self.recommender
        if self.loss_name != "bpr_loss":
            if self.world > 1:
                raise ValueError("--loss=%s runs through the generic autograd path, which is single-GPU" % self.loss_name)
        elif getattr(rec, "_lazy", False) and rec.latent_dim % (4 * self.world) != 0:
            # (the replicated-table trainer below no longer carries the folded forms a lazy model needs: fail here, by name, instead
            # of at the first step)
            ok = [w for w in range(1, 9) if rec.latent_dim % (4 * w) == 0]
            raise ValueError("recdim %d does not split into 4-float column groups over %d ranks: the column-shard engine needs "
                             "recdim %% (4 x world) == 0 -- launch on %s GPUs (or pad recdim)" % (rec.latent_dim, self.world, ok))
        elif getattr(rec, "_lazy", False):
            # the column-shard engine with this job's world / rank. It attaches itself to the model (elimrec_amd/plugin.py): the
            # epoch loop below is the reference's own four lines (main.py:98-101) and runs on it
            self.engine = ColumnShardEngine(rec)
            self.trainer = ColumnShardTrainer(self.engine, self.opt, world_size=self.world, rank=self.rank)
        else:       # adjacencies with a diagonal (norm / mean+I) without --propagation=folded, layer_num < 2: replicated tables
            self.trainer = DataParallelTrainer(rec, self.opt, world_size=self.world, rank=self.rank)
        self.start_epoch, self.resume_state = 0, None
        if "resume" in cfg and cfg["resume"]:
            self.load_checkpoint(str(cfg["resume"]))

    # ------------------------------------------------------------------ checkpoint / resume
    def save_checkpoint(self, path, epoch, extra):
        """The reference's checkpoint (state_dict, main.py:131-133) + a side file with what a resume needs on top."""
        rec = self.recommender
        if self.engine is not None:
            self.engine.sync_to_model()                       # all ranks (collective when world > 1)
            emb = self.engine.optimizer_state()
        else:
            emb = None
        if self.rank != 0:
            return
        torch.save(rec.state_dict(), path)
        torch.save(dict(epoch=epoch, embedding_adam=emb, adam=self.opt.export_state(rec.named_parameters()), extra=extra),
                   path + ".resume")

    def load_checkpoint(self, path):
        rec = self.recommender
        state = torch.load(path, map_location="cpu")
        rec.load_state_dict(state, strict=True)
        if self.engine is not None:
            self.engine.load_from_model()
        side = path + ".resume"
        if os.path.exists(side):
            extra = torch.load(side, map_location="cpu", weights_only=True)     # tensors, numbers, strings, dicts only
            rec._workspace(1)                                  # parameters move into the flat buffers before the moments
            self.opt.import_state(rec.named_parameters(), extra["adam"])
            if self.engine is not None and extra.get("embedding_adam") is not None:
                self.engine.load_optimizer_state(extra["embedding_adam"])
            self.start_epoch = int(extra["epoch"]) + 1
            self.resume_state = extra.get("extra")
            Logger.info("[resumed] %s at epoch %d (optimizer state restored)" % (path, self.start_epoch))
        else:
            Logger.info("[resumed] %s (parameters only: no %s)" % (path, side))

    # ------------------------------------------------------------------ pieces of an epoch
    def generic_step(self, users, pos, neg):
        """main.py:98-101 as written: loss method -> zero_grad -> backward -> optimizer step."""
        loss = getattr(self.recommender, self.loss_name)(users, pos, neg)
        self.opt.zero_grad()
        loss.backward(retain_graph=True)
        self.opt.step()
        return loss

    def train_epoch(self, batches):
        """One pass over the sampler; the mean loss of the epoch. One rank: the reference's line 102 verbatim --
        `loss.cpu().item()` after every batch (the engine publishes each step's loss to the host from the launch that sums it,
        elimrec_amd/plugin.py: the read does not wait for the step). Several ranks: the losses stay on the device and are
        all-reduced once per epoch."""
        self.recommender.train()
        tracker = Meter(name="MultiLoss(bpr)")
        tracker.reset()
        # the reference's loop body; only the replicated-table fallback (no engine) has a step call of its own
        step = self.generic_step if (self.trainer is None or self.engine is not None) else self.trainer.step
        # the column-shard engine returns views into a ring of loss slots: an epoch longer than the ring keeps copies
        ring = getattr(self.engine, "loss_ring_len", 0) if self.engine is not None else 0
        keep = (lambda t: t.clone()) if ring and len(batches) >= ring else (lambda t: t)
        shard_trainer = self.trainer if self.engine is not None else None
        if shard_trainer is not None:
            batches = list(batches)                   # the epoch's triplets, sampled on the device in one launch (data/sampler.py)
            shard_trainer.prestage(batches)           # ... complete before the first step: every step's planner may run ahead
            if getattr(shard_trainer, "lookup", False) and getattr(shard_trainer, "multi", False):
                shard_trainer.plan_lookup(batches)    # row-sharded constants: this epoch's lookup split sizes, planned ahead
        if self.world == 1:
            for users, pos, neg in batches:
                loss = step(users, pos, neg)
                tracker.update(val=loss.cpu().item())                               # main.py:102
        else:
            import torch.distributed as dist
            on_device = torch.stack([keep(step(users, pos, neg).detach()) for users, pos, neg in batches])
            dist.all_reduce(on_device, op=dist.ReduceOp.SUM)
            on_device /= self.world
            for value in on_device.cpu().tolist():
                tracker.update(val=value)
        if self.world > 1:       # every rank sees a bad index of any rank: all raise together, none is left in a collective
            import torch.distributed as dist
            dist.all_reduce(self.recommender._index_err(), op=dist.ReduceOp.MAX)
        self.recommender.check_indices()
        return tracker.avg

    def validate(self, epoch, meters):
        """TIE validation metrics of this epoch -> the meters; returns the (precision, recall, ndcg) triple or None."""
        rec = self.recommender
        rec.eval()
        Logger.info("[VALID]")
        Logger.info("[CF Mode]")
        for m in meters.values():
            m.reset_time()
        rec.predict_type = "TIE"
        result, _ = rec.evaluate()
        if result is not None:
            for key, value in zip(("precision", "recall", "ndcg"), result):
                meters[key].update(val=value, epoch=epoch)
            Logger.info("[{}]\t{}\t{}\t{}".format("TIE", meters["recall"], meters["ndcg"], meters["precision"]))
        return result

    def test_all_effects(self):
        """TE and TIE metrics on the test split, formatted as the reference prints them."""
        rec, lines = self.recommender, {}
        for effect in EFFECTS:
            rec.predict_type = effect
            result, _ = rec.test()
            assert result is not None
            lines[effect] = "  [{}]\t{}\t{}\t{}".format(effect, result[1], result[0], result[2])
            Logger.info(lines[effect])
        return lines

    # ------------------------------------------------------------------ the run
    def run(self):
        cfg, rec = self.config, self.recommender
        k = str(cfg["topks"][0])
        stamp = int(time.time())
        loss_meter = Meter("loss", id=stamp)
        meters = {"recall": Meter("R@" + k, id=stamp), "precision": Meter("P@" + k, id=stamp), "ndcg": Meter("N@" + k, id=stamp)}
        if cfg.batch_size % self.world != 0:
            raise ValueError("batch_size %d is not a multiple of the %d ranks" % (cfg.batch_size, self.world))
        batches = PairwiseSamplerV2(self.dataset, neg_num=1, batch_size=cfg.batch_size // self.world, shuffle=True,
                                    device=cfg.device, seed=cfg.seed + self.rank,
                                    shard=(self.rank, self.world) if self.world > 1 else None)
        batches.epoch = self.start_epoch
        best_recall = dict.fromkeys(EFFECTS, 0)
        best_epoch = dict.fromkeys(EFFECTS, 0)
        best_valid_line, test_lines = "", dict.fromkeys(EFFECTS, "")
        checkpoint = rec.getFileName()
        if self.resume_state:
            best_recall, best_epoch = dict(self.resume_state["best_recall"]), dict(self.resume_state["best_epoch"])
            best_valid_line, test_lines = self.resume_state["best_valid_line"], dict(self.resume_state["test_lines"])
        for epoch in range(self.start_epoch, cfg.num_epoch):
            Logger.info("======================")
            Logger.info("EPOCH[%d/%d]" % (epoch, cfg.num_epoch))
            loss_meter.reset_time()
            epoch_loss = self.train_epoch(batches)
            if (epoch + 1) % cfg["test_step"] == 0:
                result = self.validate(epoch, meters)
                if meters["recall"].val > best_recall["TIE"] and epoch != 0:
                    best_valid_line = "[EPOCH {}]\n{}\t{}\t{}".format(epoch, meters["recall"], meters["ndcg"], meters["precision"])
                    Logger.info("[Better Result]")
                    Logger.info("[TEST]")
                    for effect in EFFECTS:
                        best_recall[effect], best_epoch[effect] = float(result[1]), epoch
                    test_lines = self.test_all_effects()
                    if cfg["save_flag"]:
                        Logger.info("[saved][EPOCH %d]" % epoch)
                        self.save_checkpoint(checkpoint, epoch, dict(best_recall=best_recall, best_epoch=best_epoch,
                                                                     best_valid_line=best_valid_line, test_lines=test_lines))
                if epoch - best_epoch["TIE"] > cfg.stop_cnt:
                    break
            loss_meter.update(val=epoch_loss, epoch=epoch)
            Logger.info("%s" % loss_meter)
        Logger.info("=>[{}] best_valid_result:\n{}".format("TIE", best_valid_line))
        for effect in EFFECTS:
            Logger.info("=>[{}] test_result:\n{}".format(effect, test_lines[effect]))
        return best_recall, test_lines


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    os.chdir(here)
    config = Configurator(os.path.join(here, "NeuRec.properties"), default_section="hyperparameters")
    if "gpus" in config and int(config["gpus"]) > 1 and "WORLD_SIZE" not in os.environ:
        # --gpus=N from a bare shell: start one rank per GPU as child processes (this parent has not touched the GPU)
        import socket
        import subprocess
        import sys
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                          str(int(config["gpus"])), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                          os.path.abspath(__file__)] + sys.argv[1:], env=env))
    set_seed(config["seed"])
    Net(config).run()
