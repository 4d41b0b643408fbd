"""The two space types the Dart path needs, with the reference's sampling semantics
(reference gym/spaces/box.py:24-56,70-110; gym/spaces/tuple.py; gym/vector/utils/spaces.py batch_space)."""
import numpy as np

from . import seeding


class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self.np_random = None
        self.seed()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    """Box(low, high, shape=None, dtype=float32); bounded dims sample U[low, high) in float64 then cast."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            high = np.asarray(high)
            assert low.shape == high.shape, "box dimension mismatch"
            shape = low.shape
        else:
            shape = tuple(shape)
            low = np.full(shape, low) if np.isscalar(low) else np.asarray(low)
            high = np.full(shape, high) if np.isscalar(high) else np.asarray(high)
        self.low = low.astype(dtype)
        self.high = high.astype(dtype)
        self.bounded_below = -np.inf < self.low
        self.bounded_above = np.inf > self.high
        super().__init__(shape, dtype)

    def sample(self):
        out = np.empty(self.shape)
        unb = ~self.bounded_below & ~self.bounded_above
        upp = ~self.bounded_below & self.bounded_above
        low = self.bounded_below & ~self.bounded_above
        bnd = self.bounded_below & self.bounded_above
        out[unb] = self.np_random.normal(size=unb[unb].shape)
        out[low] = self.np_random.exponential(size=low[low].shape) + self.low[low]
        out[upp] = -self.np_random.exponential(size=upp[upp].shape) + self.high[upp]
        out[bnd] = self.np_random.uniform(low=self.low[bnd], high=self.high[bnd], size=bnd[bnd].shape)
        return out.astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low)) and bool(np.all(x <= self.high))

    def __repr__(self):
        return "Box" + str(self.shape)

    def __eq__(self, other):
        return isinstance(other, Box) and self.shape == other.shape and np.allclose(self.low, other.low) and \
            np.allclose(self.high, other.high)


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)
        super().__init__(None, None)

    def seed(self, seed=None):
        # the vector env's action space is Tuple((single_space,) * num_envs): one object repeated -- seeding it once leaves
        # the same final state as the reference's loop (tuple.py) and keeps a 65 536-env constructor from building 65 536
        # RandomStates (4 s)
        done, out = {}, []
        for s in self.spaces:
            if id(s) not in done:
                done[id(s)] = s.seed(seed)
            out.append(done[id(s)])
        return out

    def sample(self):
        s0 = self.spaces[0] if self.spaces else None
        if len(self.spaces) > 1 and isinstance(s0, Box) and all(s is s0 for s in self.spaces) and \
                bool(np.all(s0.bounded_below & s0.bounded_above)):
            # one bounded Box repeated (the vector env's action space): the N consecutive uniform(low, high, shape) calls of
            # the loop below consume the stream exactly like one call of shape (N,) + shape (numpy fills in C order)
            n = len(self.spaces)
            out = s0.np_random.uniform(low=s0.low, high=s0.high, size=(n,) + s0.shape).astype(s0.dtype)
            return tuple(out)
        return tuple(s.sample() for s in self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(s.contains(p) for s, p in zip(self.spaces, x))

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


def batch_space(space: Box, n: int) -> Box:
    """(k,) Box -> (n, k) Box, as gym.vector does for the batched observation space."""
    reps = (n,) + (1,) * len(space.shape)
    return Box(np.tile(space.low, reps), np.tile(space.high, reps), dtype=space.dtype)
