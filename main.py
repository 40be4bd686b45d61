#!/usr/bin/env python3
"""NeuRec-style driver with the reference's command line (reference main.py:27-175):

    python main.py --recommender=EliMRec --data.input.dataset=<name> --alpha=0.5 --loss=bpr_loss [--key=value ...]

Behaviour kept from the reference's `Net.run`: one pass of the pairwise sampler per epoch, validation every
`test_step` epochs with predict_type TIE, a checkpoint + TE/TIE test pass whenever validation recall improves
(not on epoch 0), early stop after `stop_cnt` epochs without improvement, the same log lines. The per-batch work,
the sampler and the evaluator run on the GPU (elimrec_amd). `--data.input.dataset=synthetic` uses the seeded
Tiktok-shape generator instead of reading files. Multi-GPU: launch with torch.distributed.run, one process per GPU
(elimrec_amd/dist.py).
"""
import os
import time

import torch

from elimrec_amd import (Configurator, Dataset, EliMRec, FusedAdam, Logger, Meter, PairwiseSamplerV2,
                         SyntheticDataset, set_seed)
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
    print("use", "cuda:%d" % local_rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
        self.trainer = DataParallelTrainer(self.recommender, self.opt, world_size=self.world, rank=self.rank)

    # ------------------------------------------------------------------ pieces of an epoch
    def train_epoch(self, batches):
        """One pass over the sampler; the mean loss of the epoch (one device->host copy for all batches instead of the
        reference's per-batch loss.cpu().item())."""
        self.recommender.train()
        tracker = Meter(name="MultiLoss(bpr)")
        tracker.reset()
        on_device = [self.trainer.step(users, pos, neg) for users, pos, neg in batches]
        for value in torch.stack(on_device).cpu().tolist():
            tracker.update(val=value)
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
        batches = PairwiseSamplerV2(self.dataset, neg_num=1, batch_size=cfg.batch_size, shuffle=True, device=cfg.device,
                                    seed=cfg.seed + self.rank)
        best_recall = dict.fromkeys(EFFECTS, 0)
        best_epoch = dict.fromkeys(EFFECTS, 0)
        best_valid_line, test_lines = "", dict.fromkeys(EFFECTS, "")
        checkpoint = rec.getFileName()
        for epoch in range(cfg.num_epoch):
            Logger.info("======================")
            Logger.info("EPOCH[%d/%d]" % (epoch, cfg.num_epoch))
            loss_meter.reset_time()
            epoch_loss = self.train_epoch(batches)
            if (epoch + 1) % cfg["test_step"] == 0:
                result = self.validate(epoch, meters)
                if meters["recall"].val > best_recall["TIE"] and epoch != 0:
                    if cfg["save_flag"] and self.rank == 0:
                        Logger.info("[saved][EPOCH %d]" % epoch)
                        torch.save(rec.state_dict(), checkpoint)
                    best_valid_line = "[EPOCH {}]\n{}\t{}\t{}".format(epoch, meters["recall"], meters["ndcg"], meters["precision"])
                    Logger.info("[Better Result]")
                    Logger.info("[TEST]")
                    for effect in EFFECTS:
                        best_recall[effect], best_epoch[effect] = result[1], epoch
                    test_lines = self.test_all_effects()
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
    set_seed(config["seed"])
    Net(config).run()
