"""Wall-clock context manager (same contract as numbskull/timer.py:7-18: ``.interval`` in
seconds after the block)."""

import time


class Timer:
    def __enter__(self):
        self.start = time.time()
        return self

    def __exit__(self, *exc):
        self.end = time.time()
        self.interval = self.end - self.start
