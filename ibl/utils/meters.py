class AverageMeter(object):
    """Last value and running mean of a stream of measurements (`val` / `avg`, with `sum` and
    `count` exposed because callers print them); `update(v, n)` counts `v` as n samples."""

    __slots__ = ("val", "sum", "count")

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.sum, self.count = 0, 0, 0

    def update(self, val, n=1):
        self.val = val
        self.sum = self.sum + val * n
        self.count = self.count + n

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0
