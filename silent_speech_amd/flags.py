"""Hyper-parameters of the reference's absl FLAGS (defined at import time across architecture.py:10-12,
transduction_model.py:22-31, recognition_model.py:20-28).  If absl is importable and the flag is
defined there (i.e. we are imported next to the reference's entry points) its value wins; otherwise
these defaults -- identical to the reference's transduction trainer -- are used (the recognition trainer
passes its own defaults through FLAGS.lookup).  Everything also takes explicit arguments."""

DEFAULTS = dict(model_size=768, num_layers=6, dropout=0.2,                       # architecture.py:10-12
                batch_size=32, epochs=80, learning_rate=1e-3, learning_rate_patience=5,
                learning_rate_warmup=500, start_training_from=None, data_size_fraction=1.0,
                phoneme_loss_weight=0.5, l2=1e-7, output_directory='output')      # transduction_model.py:22-31


class _Flags(object):
    def __init__(self):
        object.__setattr__(self, '_over', {})

    def __getattr__(self, k):
        over = object.__getattribute__(self, '_over')
        if k in over:
            return over[k]
        try:
            from absl import flags as _af
            try:
                return getattr(_af.FLAGS, k)
            except Exception:
                pass
        except Exception:
            pass
        if k in DEFAULTS:
            return DEFAULTS[k]
        raise AttributeError(k)

    def lookup(self, k, fallback):
        """Like attribute access, but with the caller's default (recognition_model.py defines the same flag names
        with different defaults than transduction_model.py)."""
        over = object.__getattribute__(self, '_over')
        if k in over:
            return over[k]
        try:
            from absl import flags as _af
            return getattr(_af.FLAGS, k)
        except Exception:
            return fallback

    def __setattr__(self, k, v):
        self._over[k] = v


FLAGS = _Flags()
