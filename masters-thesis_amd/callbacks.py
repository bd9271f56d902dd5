"""Keras callback protocol objects used by AttemptFour/main.py:197-232 (SURVEY component 12).

``ModelBase.fit`` calls ``on_train_begin, on_epoch_begin, on_train_batch_begin,
on_train_batch_end(batch, logs), on_test_batch_end(batch, logs), on_epoch_end(epoch, logs),
on_train_end`` and sets ``.model``; any object with these methods works (the reference's
``Callbacks/EpochLoss.py`` LossHistory included, once its keras base class is swapped for
``Callback`` below).
"""
import csv
import os


class Callback:
    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model


class LossHistory(Callback):
    """Callbacks/EpochLoss.py:12-52: per-batch training/validation logs -> CSV at epoch end."""

    def __init__(self, file_name, out_dir=None):
        super().__init__()
        self.file_name, self.rows, self.epoch = file_name, [], 0

    def on_epoch_begin(self, epoch, logs=None):
        self.epoch = epoch

    def on_train_batch_end(self, batch, logs=None):
        self.rows.append(dict(epoch=self.epoch, batch=batch, phase="train", **(logs or {})))

    def on_test_batch_end(self, batch, logs=None):
        self.rows.append(dict(epoch=self.epoch, batch=batch, phase="val", **{f"val_{k}": v for k, v in (logs or {}).items()}))

    def on_epoch_end(self, epoch, logs=None):
        keys = []
        for r in self.rows:
            for k in r:
                if k not in keys:
                    keys.append(k)
        with open(self.file_name, "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            w.writerows(self.rows)


class LearningRateScheduler(Callback):
    """tf.keras.callbacks.LearningRateScheduler(schedule) -- main.py:86-92,218-219."""

    def __init__(self, schedule, verbose=0):
        super().__init__()
        self.schedule = schedule

    def on_epoch_begin(self, epoch, logs=None):
        self.model.optimizer.lr = float(self.schedule(epoch))


class ModelCheckpoint(Callback):
    """tf.keras.callbacks.ModelCheckpoint(filepath, monitor, save_weights_only=True, save_best_only, mode)
    -- main.py:168-190.  Weights are written by ``model.save_weights`` (.npz, keras names/layouts)."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_weights_only=True, save_best_only=False, mode="min",
                 period=1):
        super().__init__()
        self.filepath, self.monitor, self.best_only, self.sign = filepath, monitor, save_best_only, (1 if mode == "min" else -1)
        self.best = float("inf")

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        cur = logs.get(self.monitor)
        rank0 = True
        if getattr(self.model, "dp_world", 1) > 1:
            # data parallel: every rank averages the per-replica BatchNorm moving statistics (a collective,
            # so it runs before any rank-local decision), then only rank 0 writes the file
            import torch.distributed as dist
            from . import dp
            dp.average_moving_stats(self.model)
            rank0 = dist.get_rank() == 0
        if not rank0:
            return
        if self.best_only:
            if cur is None or self.sign * cur >= self.best:
                return
            self.best = self.sign * cur
        path = self.filepath.format(epoch=epoch + 1, **logs)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        self.model.save_weights(path)


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0.0, patience=0):
        super().__init__()
        self.monitor, self.min_delta, self.patience = monitor, min_delta, patience
        self.best, self.wait = float("inf"), 0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait > self.patience:
                self.model.stop_training = True
