#!/usr/bin/env python3
"""NeuRec-style driver with the reference's command line (main.py:27-175):

    python main.py --recommender=EliMRec --data.input.dataset=<name> --alpha=0.5 --loss=bpr_loss [--key=value ...]

Same epoch loop, evaluation cadence (`test_step`), best-checkpoint rule, early stopping
(`stop_cnt`) and log lines as the reference's `Net.run`; the per-batch work, the sampler and the
evaluator run on the GPU (elimrec_amd). `--data.input.dataset=synthetic` uses the seeded
Tiktok-shape generator instead of reading files. Multi-GPU: launch with torch.distributed.run,
one process per GPU (elimrec_amd/dist.py).
"""
import os
import sys
import time

import torch

from elimrec_amd import (Configurator, Dataset, EliMRec, FusedAdam, Logger, Meter, PairwiseSamplerV2,
                         SyntheticDataset, set_seed)
from elimrec_amd.dist import DataParallelTrainer


def get_parameter_number(net):
    total = sum(p.numel() for p in net.parameters())
    return {"Total": total, "Trainable": sum(p.numel() for p in net.parameters() if p.requires_grad)}


class Net(object):
    def __init__(self, args):
        self.config = args
        cfg = args
        Logger.logger = Logger(name="_".join([cfg.recommender, cfg["data.input.dataset"], cfg.loss, cfg.suffix]),
                               show_in_console=cfg.verbose != 0, is_creat_log_file=cfg.create_log_file,
                               path=cfg.log_path)
        Logger.info(cfg.params_str())
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        use_gpu = torch.cuda.is_available() and not cfg.no_cuda
        print("use", "cuda:%d" % local_rank if use_gpu else "cpu")
        if not use_gpu:
            raise SystemExit("elimrec_amd has no CPU path for training: an MI355X is required (no_cuda must be FALSE)")
        torch.cuda.set_device(local_rank)
        cfg.device = torch.device("cuda", local_rank)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=cfg.device)
        if cfg["data.input.dataset"] == "synthetic":
            shape = cfg["synthetic_shape"] if "synthetic_shape" in cfg else [36656, 76085, 720829]
            dims = cfg["synthetic_dims"] if "synthetic_dims" in cfg else [128, 128, 128]
            self.dataset = SyntheticDataset(shape[0], shape[1], shape[2], feat_dims=tuple(dims), seed=0)
        else:
            self.dataset = Dataset(cfg)
        if cfg.recommender != "EliMRec":
            raise ValueError("unknown recommender '%s'" % cfg.recommender)
        self.recommender = EliMRec(cfg, self.dataset).to(cfg.device)
        self.cf_mode = True if "cf_mode" not in cfg else cfg["cf_mode"]
        Logger.info(get_parameter_number(self.recommender))
        self.opt = FusedAdam(self.recommender.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
        self.trainer = DataParallelTrainer(self.recommender, self.opt, world_size=self.world, rank=self.rank)

    def run(self):
        cfg, rec = self.config, self.recommender
        topk = str(cfg["topks"][0])
        meter_id = int(time.time())
        loss_meter = Meter("loss", id=meter_id)
        data_iter = PairwiseSamplerV2(self.dataset, neg_num=1, batch_size=cfg.batch_size, shuffle=True,
                                      device=cfg.device, seed=cfg.seed + self.rank)
        best_recall = {"TE": 0, "TIE": 0}
        best_epoch = {"TE": 0, "TIE": 0}
        best_result = {"TE": "", "TIE": ""}
        final_result = {"TE": "", "TIE": ""}
        recall_meter = {"TIE": Meter("R@" + topk, id=meter_id)}
        precision_meter = {"TIE": Meter("P@" + topk, id=meter_id)}
        ndcg_meter = {"TIE": Meter("N@" + topk, id=meter_id)}
        model_name = rec.getFileName()
        for epoch in range(cfg.num_epoch):
            rec.train()
            Logger.info("======================")
            Logger.info("EPOCH[%d/%d]" % (epoch, cfg.num_epoch))
            loss_meter.reset_time()
            batch_loss_meter = Meter(name="MultiLoss(bpr)")
            batch_loss_meter.reset()
            losses = []
            for bat_users, bat_pos_items, bat_neg_items in data_iter:
                batch_loss_meter.reset_time()
                losses.append(self.trainer.step(bat_users, bat_pos_items, bat_neg_items))
            # one device->host copy per epoch instead of the reference's per-batch loss.cpu().item()
            for v in torch.stack(losses).cpu().tolist():
                batch_loss_meter.update(val=v)
            if (epoch + 1) % cfg["test_step"] == 0:
                Logger.info("[VALID]")
                rec.eval()
                Logger.info("[CF Mode]")
                effect = "TIE"
                for m in (recall_meter, precision_meter, ndcg_meter):
                    m[effect].reset_time()
                rec.predict_type = effect
                current_result, buf = rec.evaluate()
                if current_result is not None:
                    recall_meter[effect].update(val=current_result[1], epoch=epoch)
                    precision_meter[effect].update(val=current_result[0], epoch=epoch)
                    ndcg_meter[effect].update(val=current_result[2], epoch=epoch)
                    Logger.info("[{}]\t{}\t{}\t{}".format(effect, recall_meter[effect], ndcg_meter[effect],
                                                          precision_meter[effect]))
                if recall_meter["TIE"].val > best_recall["TIE"] and epoch != 0:
                    if cfg["save_flag"] and self.rank == 0:
                        Logger.info("[saved][EPOCH %d]" % epoch)
                        torch.save(rec.state_dict(), model_name)
                    best_result["TIE"] = "[EPOCH {}]\n{}\t{}\t{}".format(epoch, recall_meter[effect], ndcg_meter[effect],
                                                                         precision_meter[effect])
                    Logger.info("[Better Result]")
                    Logger.info("[TEST]")
                    for effect in ["TE", "TIE"]:
                        best_recall[effect] = current_result[1]
                        best_epoch[effect] = epoch
                        rec.predict_type = effect
                        test_result, _ = rec.test()
                        assert test_result is not None
                        final_result[effect] = "  [{}]\t{}\t{}\t{}".format(effect, test_result[1], test_result[0],
                                                                           test_result[2])
                        Logger.info(final_result[effect])
                if (epoch - best_epoch["TIE"]) > cfg.stop_cnt:
                    break
            loss_meter.update(val=batch_loss_meter.avg, epoch=epoch)
            Logger.info("%s" % loss_meter)
        for effect in ["TIE"]:
            Logger.info("=>[{}] best_valid_result:\n{}".format(effect, best_result[effect]))
        for effect in ["TE", "TIE"]:
            Logger.info("=>[{}] test_result:\n{}".format(effect, final_result[effect]))
        return best_recall, final_result


if __name__ == "__main__":
    root_folder = os.path.dirname(os.path.abspath(__file__))
    os.chdir(root_folder)
    args = Configurator(os.path.join(root_folder, "NeuRec.properties"), default_section="hyperparameters")
    set_seed(args["seed"])
    Net(args).run()
