"""Running mean of a logged quantity with the attribute interface the trainers' progress lines read
(`.val`, `.avg`, `.sum`, `.count`, `reset()`, `update(value, n)`; CC/clustercontrast/utils/meters.py).  Host-side
bookkeeping only."""
from __future__ import absolute_import


class AverageMeter(object):
    __slots__ = ("val", "sum", "count")

    def __init__(self):
        self.val, self.sum, self.count = 0, 0, 0

    reset = __init__

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0

    def update(self, value, n=1):
        self.val = value
        self.sum = self.sum + value * n
        self.count = self.count + n
