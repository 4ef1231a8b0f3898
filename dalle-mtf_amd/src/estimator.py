"""Minimal Estimator: the loop the reference delegates to tf TPUEstimator (train_dalle.py:71-98,
train_vae_tf.py:63-95): build once via model_fn, iterate input_fn batches, run train_op, step-numbered
checkpoints under model_dir with max_to_keep, resume from the latest (src/model_fns.py:204-229)."""
import glob
import os
import re
import time
from collections import namedtuple

import torch

from .utils import ModeKeys

EstimatorSpec = namedtuple("EstimatorSpec", ["mode", "loss", "train_op", "host_call", "training_hooks", "eval_metrics"])
EstimatorSpec.__new__.__defaults__ = (None, None, None, None, None)


def latest_checkpoint(model_dir):
    """tf.train.latest_checkpoint analogue: newest model.ckpt-<step>.pt under model_dir."""
    if not model_dir or not os.path.isdir(model_dir):
        return None
    best, best_step = None, -1
    for p in glob.glob(os.path.join(model_dir, "model.ckpt-*.pt")):
        m = re.search(r"model\.ckpt-(\d+)\.pt$", p)
        if m and int(m.group(1)) > best_step:
            best, best_step = p, int(m.group(1))
    return best


def load_global_step_from_checkpoint_dir(model_dir):
    """estimator_lib._load_global_step_from_checkpoint_dir (train_dalle.py:39)."""
    p = latest_checkpoint(model_dir)
    if p is None:
        return 0
    return int(re.search(r"model\.ckpt-(\d+)\.pt$", p).group(1))


class CheckpointSaverHook:
    """tf.train.CheckpointSaverHook(save_steps) + Saver(max_to_keep) (src/model_fns.py:209-222)."""

    def __init__(self, model_dir, save_steps, get_state, max_to_keep=5, is_chief=True):
        self.model_dir, self.save_steps, self.get_state = model_dir, save_steps, get_state
        self.max_to_keep, self.is_chief = max_to_keep, is_chief

    def save(self, step):
        if not self.model_dir:
            return
        if not self.is_chief:
            self._sync()
            return
        err, interrupt = None, None
        try:
            os.makedirs(self.model_dir, exist_ok=True)
            path = os.path.join(self.model_dir, f"model.ckpt-{step}.pt")
            torch.save(self.get_state(), path + ".tmp")
            os.replace(path + ".tmp", path)
        except Exception as e:                       # the other ranks are waiting in _sync(): reach it, then fail everywhere
            err = e
        except (KeyboardInterrupt, SystemExit) as e:   # same, but the chief re-raises the interrupt as itself afterwards
            err = interrupt = e
        if err is None:
            # pruning old checkpoints happens after the new one is committed: a failure here is a warning, not a failed save
            # (any failure: the other ranks wait in _sync() below and must be reached -- a stray "model.ckpt-best.pt" in the
            # directory is skipped by the match, anything else that goes wrong here is reported and ignored)
            try:
                ck = [(int(m.group(1)), p) for p in glob.glob(os.path.join(self.model_dir, "model.ckpt-*.pt"))
                      for m in [re.search(r"model\.ckpt-(\d+)\.pt$", p)] if m]
                ck = [p for _, p in sorted(ck)]
                for old in ck[:-self.max_to_keep] if self.max_to_keep else []:
                    os.remove(old)
            except Exception as e:
                import warnings
                warnings.warn(f"checkpoint retention: could not remove an old checkpoint ({e!r}); the new one is saved")
        try:
            self._sync(err)
        finally:
            if interrupt is not None:
                raise interrupt

    def _sync(self, err=None):
        """every rank leaves save() only after the chief's file is complete (readers: evaluate(), resume); a failed write
        on the chief (disk full, permissions) raises on EVERY rank instead of leaving the others in the barrier"""
        from .dp import agree_ok
        if not agree_ok(err is None):
            raise RuntimeError(f"checkpoint save failed on the chief rank: {err!r}") from err

    def after_step(self, step):
        if self.save_steps and step % self.save_steps == 0:
            self.save(step)


class Estimator:
    def __init__(self, model_fn, model_dir, params, log_every=100, logger=None):
        self.model_fn, self.model_dir, self.params = model_fn, model_dir, params
        self.log_every, self.logger = log_every, logger
        self._train_it = None     # ONE input stream for the life of the estimator: train() is called once per
                                  # steps_per_checkpoint segment when eval_steps > 0 and must not replay the stream's head

    def _log(self, msg):
        (self.logger.info if self.logger else print)(msg)

    def train(self, input_fn, max_steps):
        params = self.params
        if self._train_it is None:
            # a resumed run continues a different stream than the one that produced the checkpoint: the start step salts
            # the shuffle seeds (the reference's tf.data shuffle is unseeded and never replays either)
            params["_input_start_step"] = load_global_step_from_checkpoint_dir(self.model_dir)
            self._train_it = iter(input_fn(params))
        it = self._train_it
        t0, n0 = time.time(), None
        spec = None
        while True:
            features, labels = next(it)
            spec = self.model_fn(features, labels, ModeKeys.TRAIN, params)
            step = spec.train_op()
            if n0 is None:
                n0 = step
            for hk in spec.training_hooks or []:
                hk.after_step(step)
            if step % self.log_every == 0:
                dt = time.time() - t0
                self._log(f"step {step}: loss {float(spec.loss):.4f}  ({(step - n0 + 1) / max(dt, 1e-9):.2f} steps/s)")
                if spec.host_call:
                    fn, tensors = spec.host_call
                    fn(step, **tensors)
            if step >= max_steps:
                break
        for hk in spec.training_hooks or []:
            if isinstance(hk, CheckpointSaverHook):
                hk.save(step)
        return step

    def close(self):
        """stop and JOIN the input producer (src/input_fns._Prefetch.close): every entry point calls this in a `finally`
        (or uses the estimator as a context manager) so no producer thread is alive at interpreter exit"""
        if self._train_it is not None and hasattr(self._train_it, "close"):
            self._train_it.close()
        self._train_it = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def evaluate(self, input_fn, steps):
        it = iter(input_fn(self.params))
        tot = 0.0
        try:
            for _ in range(steps):
                features, labels = next(it)
                spec = self.model_fn(features, labels, ModeKeys.EVAL, self.params)
                tot += float(spec.loss)
        finally:
            if hasattr(it, "close"):
                it.close()          # joins the eval producer
        self._log(f"eval: mean loss {tot / max(steps, 1):.4f} over {steps} steps")
        return {"loss": tot / max(steps, 1)}
