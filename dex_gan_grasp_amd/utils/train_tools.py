"""Mirror of DexGanGrasp/utils/train_tools.py:4-42 (EarlyStopping; instantiated by the wrapper at
dexgangrasp.py:117-118, never consulted by train.py)."""
import numpy as np


class EarlyStopping:
    def __init__(self, patience=5, min_delta=0.0):
        self.patience = patience
        self.min_delta = min_delta
        self.counter = 0
        self.best = np.inf
        self.early_stop = False

    def __call__(self, val_loss):
        val_loss = float(val_loss)
        if val_loss < self.best - self.min_delta:
            self.best = val_loss
            self.counter = 0
        else:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        return self.early_stop
