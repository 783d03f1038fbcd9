"""Container-only stand-in so the read-only Python reference imports here.

`cachetools` is not installed and cannot be fetched offline.  The reference only *uses*
LRUCache when use_hash_table=True, which no golden case exercises.  This file is used by
tools/make_golden.py only; it never travels into the product or the oracle.
"""


class LRUCache(dict):
    def __init__(self, maxsize=128):
        super().__init__()
        self.maxsize = maxsize
