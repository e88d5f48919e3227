"""Embedding-table sizes of the three datasets (reference: nasrec/utils/config.py:17-41), restated as data.

MAX_NUM_EMBEDDINGS caps every table: 0.5 M during search, uncapped for final training (reference README note)."""
MAX_NUM_EMBEDDINGS = 500000 * 10000

_CRITEO = [1461, 584, 10131227, 2202609, 306, 25, 12518, 634, 4, 93146, 5684, 8351593, 3195, 28, 14993, 5461307, 11, 5653, 2174, 5,
           7046548, 19, 16, 286182, 106, 142573]
_AVAZU = [10000, 241, 8, 8, 4738, 7746, 27, 8553, 560, 37, 2686409, 6729487, 8252, 6, 5, 2627, 9, 10, 436, 5, 69, 173, 61]
_KDD = [26274, 641708, 14848, 22122011, 1188090, 3735797, 2934102, 20004011, 4, 8]


def capped(sizes, cap=MAX_NUM_EMBEDDINGS):
    return [min(x, cap) for x in sizes]


NUM_EMBEDDINGS_CRITEO = capped(_CRITEO)
NUM_EMBEDDINGS_AVAZU = capped(_AVAZU)
NUM_EMBEDDINGS_KDD = capped(_KDD)
NUM_EMBEDDINGS_TEST = [100] * 26

DATASETS = {
    "criteo": dict(Fd=13, Fs=26, tables=_CRITEO),
    "avazu": dict(Fd=1, Fs=23, tables=_AVAZU),
    "kdd": dict(Fd=3, Fs=10, tables=_KDD),
}
