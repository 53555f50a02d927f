"""Training constants of the reference (lib/config.py:62-71)."""


class _NS(object):
    def __init__(self, **kw):
        self.__dict__.update(kw)


CONF = _NS(
    TRAIN=_NS(MAX_DES_LEN=30, SEED=42, OVERLAID_THRESHOLD=0.5,
              MIN_IOU_THRESHOLD=0.25, NUM_BINS=6),
    EVAL=_NS(MIN_IOU_THRESHOLD=0.5),
)
