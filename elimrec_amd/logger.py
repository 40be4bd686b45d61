"""Console/file logger and running-average meter with the reference's line formats
(util/logger.py:4-39, util/meter.py:5-59)."""
import os
import sys
import time


def _emit(text):
    """One console line as ONE write + flush: the ranks of a multi-rank job share a pipe, and `print` issues the text and the
    newline as separate writes that another rank's line can land between. In a job of several ranks (WORLD_SIZE > 1) every
    line carries `[rank k]`; a one-rank run prints the reference's lines unchanged (util/logger.py:24-30)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        tag = "[rank %s]\t" % os.environ.get("RANK", "0")
        text = "\n".join(tag + l for l in text.split("\n"))
    sys.stdout.write(text + "\n")
    sys.stdout.flush()


class Logger(object):
    logger = None   # process-wide instance, set by the driver (main.py:29-33)

    def __init__(self, name=None, show_in_console=False, is_creat_log_file=False, path=""):
        stamp = str(int(time.time()))
        self.name = stamp if name is None else name + "_" + stamp
        self.show_in_console = show_in_console
        self.in_file = is_creat_log_file
        self.log_file_name = path + "/" + self.name + ".log"
        if self.in_file is True:
            os.makedirs(path, exist_ok=True)
            with open(self.log_file_name, "w", encoding="utf-8") as f:
                f.write("============Start Logging============\n")
                f.write("[Created At]:" + time.asctime(time.localtime(time.time())) + "\n")
                f.write("=====================================\n")
            _emit("log to file:  " + self.log_file_name)

    def log(self, *msg):
        line = "\t".join(str(m) for m in msg)
        if self.show_in_console:
            _emit(line)
        if self.in_file is True:
            with open(self.log_file_name, "a") as f:
                f.write(line + "\n")

    @staticmethod
    def info(*msg):
        if Logger.logger is None:
            _emit(" ".join(str(m) for m in msg))
        else:
            Logger.logger.log(*msg)


HISTORY_FIELDS = ("epoch", "val", "sum", "avg", "time", "count")


class Meter(object):
    """Last value, running mean and the full update history of one scalar (a loss, a metric). Printed as
    `[name val (avg) time:t]`, t = seconds since the last reset_time() -- the line format of the reference's meters
    (util/meter.py), which the driver's log lines embed."""

    def __init__(self, name, fmt=":f", id=None):
        self.name, self.id = name, id
        self._pattern = "[%s {0%s} ({1%s}) time:{2}]" % ("{3}", fmt, fmt)
        self.reset()

    def reset(self):
        self.history = []                 # one dict per update(), keys HISTORY_FIELDS
        self.val = self.sum = self.count = self.epoch = 0
        self.time = 0
        self._t0 = 0
        return self

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0

    def reset_time(self):
        self._t0 = time.time()

    def update(self, val, epoch=0, n=1):
        self.val, self.epoch = val, epoch
        self.sum += n * val
        self.count += n
        self.time = time.time() - self._t0
        self.history.append(dict(zip(HISTORY_FIELDS, (epoch, val, self.sum, self.avg, self.time, self.count))))

    def save_history(self, path=""):
        """CSV of the history, `<path>/<name>_<id>.csv`; False when nothing was recorded."""
        if not self.history:
            print("No %s data updated!" % self.name)
            return False
        target = os.path.join(path, "%s_%s.csv" % (self.name, self.id)) if path else "/%s_%s.csv" % (self.name, self.id)
        with open(target, "w", encoding="utf-8") as out:
            out.write(",".join(HISTORY_FIELDS) + "\n")
            out.writelines(",".join(str(row[f]) for f in HISTORY_FIELDS) + "\n" for row in self.history)
        print("[saved %s %d]:to %s" % (self.name, int(time.time()), target))
        return True

    def __str__(self):
        return self._pattern.format(self.val, self.avg, self.time, self.name)
