"""Training-loop policies used by train.py (keras.callbacks look-alikes): Callback base, ModelCheckpoint
(best val_loss, weights only; train.py:194-195) and the reference's per-iteration EarlyStoppingIter
(utils.py:535-614: every `patience` batches compare the cumulative-mean loss with the best one so far; on no
improvement stop and optionally restore the best weights)."""
import warnings

import numpy as np


class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor='val_loss', verbose=0, save_best_only=False, save_weights_only=False, mode='auto', period=1):
        super().__init__()
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.save_weights_only = save_best_only, save_weights_only
        self.best = -np.inf if (mode == 'max' or (mode == 'auto' and 'acc' in monitor)) else np.inf
        self.better = np.greater if self.best == -np.inf else np.less

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None:
                warnings.warn('Can save best model only with %s available, skipping.' % self.monitor, RuntimeWarning)
                return
            if not self.better(cur, self.best):
                return
            if self.verbose:
                print('\nEpoch %05d: %s improved from %0.5f to %0.5f, saving model to %s' % (epoch + 1, self.monitor, self.best, cur, path))
            self.best = cur
        (self.model.save_weights if self.save_weights_only else self.model.save)(path)


class EarlyStoppingIter(Callback):
    def __init__(self, monitor='loss', min_delta=0, patience=5000, verbose=0, mode='auto', baseline=None, restore_best_weights=False):
        super().__init__()
        self.monitor, self.baseline, self.patience, self.verbose = monitor, baseline, patience, verbose
        self.restore_best_weights = restore_best_weights
        self.stopped_iter = 0
        self.cycle_iterations = 0
        self.best_weights = None
        self.sum_monitor = 0
        if mode not in ('auto', 'min', 'max'):
            warnings.warn('\nEarlyStopping mode %s is unknown, fallback to auto mode.' % mode, RuntimeWarning)
            mode = 'auto'
        maximise = mode == 'max' or (mode == 'auto' and 'acc' in monitor)
        self.monitor_op = np.greater if maximise else np.less
        self.min_delta = min_delta if maximise else -min_delta

    def on_train_begin(self, logs=None):
        self.stopped_iter = 0
        if self.baseline is not None:
            self.best = self.baseline
        else:
            self.best = np.inf if self.monitor_op == np.less else -np.inf

    def on_batch_end(self, batch, logs=None):
        self.cycle_iterations += 1
        logs = logs or {}
        if self.monitor not in logs:
            return
        self.sum_monitor += logs[self.monitor]
        if (self.cycle_iterations - 1) % self.patience:
            return
        current = self.sum_monitor / self.cycle_iterations
        if self.monitor_op(current - self.min_delta, self.best):
            self.best = current
            if self.restore_best_weights:
                self.best_weights = self.model.get_weights()
            return
        self.stopped_iter = self.cycle_iterations
        self.model.stop_training = True
        if self.restore_best_weights:
            if self.verbose > 0:
                print('\nRestoring model weights from the end of the best epoch')
            self.model.set_weights(self.best_weights)

    def on_train_end(self, logs=None):
        if self.stopped_iter > 0 and self.verbose > 0:
            print('\nIteration %i: early stopping\nBest metric value: %.4f' % (self.stopped_iter + 1, self.best))
