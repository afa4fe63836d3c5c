"""AverageMeter — same interface as CC/clustercontrast/utils/meters.py (val / avg / sum / count, reset, update)."""
from __future__ import absolute_import


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
