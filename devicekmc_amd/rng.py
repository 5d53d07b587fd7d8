"""Host random numbers that reproduce the reference's ``RandomNumberGenerator``.

random_num.h:4-23 wraps ``std::mt19937`` and draws through
``std::uniform_real_distribution<double>(0.0, 1.0)``.  libstdc++ builds each double from two
32-bit outputs: ``(x1 + x2 * 2**32) / 2**64`` with the sum rounded to double (``generate_canonical``),
mapping an exact 1.0 to the largest double below it.  Event selection is decided by these numbers,
so the stream has to match bit for bit.
"""
import copy

import numpy as np


class StdMT19937:
    def __init__(self, seed: int = 0):
        self.seed = int(seed)
        self.n_raw = 0                            # 32-bit outputs drawn so far: the stream position a restart file carries
        self._bg = np.random.MT19937()
        self._bg._legacy_seeding(int(seed))       # init_genrand == std::mt19937::seed(value)

    def copy(self) -> "StdMT19937":
        c = StdMT19937.__new__(StdMT19937)
        c.seed, c.n_raw = self.seed, self.n_raw
        c._bg = np.random.MT19937()
        c._bg.state = copy.deepcopy(self._bg.state)
        return c

    @classmethod
    def at_position(cls, seed: int, n_raw: int) -> "StdMT19937":
        """The generator after n_raw 32-bit outputs of std::mt19937(seed) (restart)."""
        g = cls(seed)
        left = int(n_raw)
        while left > 0:
            k = min(left, 1 << 20)
            g.raw(k); left -= k
        return g

    def raw(self, n: int) -> np.ndarray:
        self.n_raw += int(n)
        return self._bg.random_raw(n).astype(np.uint64)

    def uniform_batch(self, n: int) -> np.ndarray:
        r = self.raw(2 * n)
        s = r[0::2].astype(np.float64)
        s = s + r[1::2].astype(np.float64) * 4294967296.0
        u = s / 18446744073709551616.0
        u[u >= 1.0] = np.nextafter(1.0, 0.0)
        return u

    def uniform(self) -> float:
        return float(self.uniform_batch(1)[0])

    def skip(self, n_uniform: int) -> None:
        if n_uniform > 0:
            self.n_raw += 2 * int(n_uniform)
            self._bg.random_raw(2 * n_uniform)
