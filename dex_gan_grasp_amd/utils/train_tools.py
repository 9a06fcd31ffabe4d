"""Mirror of DexGanGrasp/utils/train_tools.py:4-42 (EarlyStopping; instantiated by the wrapper at
dexgangrasp.py:117-118, never consulted by train.py).  Same constructor defaults, attributes and stopping
rule as the reference: the monitored score is -val_loss; a call whose score stays below best_score + delta
counts against `patience`, anything else becomes the new best and resets the counter."""
import numpy as np


class EarlyStopping:
    def __init__(self, patience=7, verbose=True, delta=0, trace_func=None):
        self.patience, self.verbose, self.delta, self.trace_func = patience, verbose, delta, trace_func
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf   # the reference spells it np.Inf (removed in NumPy 2)

    def __call__(self, val_loss):
        score = -val_loss
        if self.best_score is not None and score < self.best_score + self.delta:
            self.counter += 1
            if self.trace_func is not None:
                self.trace_func('EarlyStopping counter: %d out of %d' % (self.counter, self.patience))
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            first = self.best_score is None
            self.best_score = score
            if not first:
                self.counter = 0
        return self.early_stop
