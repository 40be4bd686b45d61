"""Console/file logger and running-average meter with the reference's line formats
(util/logger.py:4-39, util/meter.py:5-59)."""
import os
import time


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
            print("log to file: ", self.log_file_name)

    def log(self, *msg):
        line = "\t".join(str(m) for m in msg)
        if self.show_in_console:
            print(line)
        if self.in_file is True:
            with open(self.log_file_name, "a") as f:
                f.write(line + "\n")

    @staticmethod
    def info(*msg):
        if Logger.logger is None:
            print(*msg)
        else:
            Logger.logger.log(*msg)


class Meter(object):
    """val / running avg / history, printed as `[name val (avg) time:t]`."""

    def __init__(self, name, fmt=":f", id=None):
        self.name, self.fmt, self.id = name, fmt, id
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0
        self.history = []
        self.epoch = 0
        self._start_time = 0
        self.time = 0
        return self

    def reset_time(self):
        self._start_time = time.time()

    def update(self, val, epoch=0, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
        self.epoch = epoch
        self.time = time.time() - self._start_time
        self.history.append(dict(epoch=epoch, val=val, sum=self.sum, avg=self.avg, time=self.time, count=self.count))

    def save_history(self, path=""):
        if not self.history:
            print("No %s data updated!" % self.name)
            return False
        file_path = "%s/%s_%s.csv" % (path, self.name, self.id)
        keys = list(self.history[0].keys())
        with open(file_path, "w", encoding="utf-8") as f:
            f.write(",".join(keys) + "\n")
            for row in self.history:
                f.write(",".join(str(row[k]) for k in keys) + "\n")
        print("[saved %s %d]:to %s" % (self.name, int(time.time()), file_path))
        return True

    def __str__(self):
        return ("[{name} {val" + self.fmt + "} ({avg" + self.fmt + "}) time:{time}]").format(**self.__dict__)
