"""Meters and the Noam schedule with the reference's behaviour (reference utils/utils.py:8-53)."""


class AverageMeter(object):
    """Running value / average with the reference's print format."""

    def __init__(self, name, fmt=':f'):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ('{name} {val' + self.fmt + '} ({avg' + self.fmt + '})').format(**self.__dict__)


class ProgressMeter(object):
    def __init__(self, num_batches, *meters, prefix=""):
        digits = len(str(num_batches // 1))
        self.batch_fmtstr = '[{:' + str(digits) + 'd}/' + ('{:' + str(digits) + 'd}').format(num_batches) + ']'
        self.meters, self.prefix = meters, prefix

    def print(self, batch):
        print('\t'.join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters]),
              flush=True)


def noam_decay(step, warmup_steps, base_lr):
    """lr = base * min(step^-0.5, step * warmup^-1.5) -- no d_model factor (utils/utils.py:49-53)."""
    return base_lr * min(step ** (-0.5), step * warmup_steps ** (-1.5))
