"""Mirror of the reference's triarray.TriArr (triarray.py:13-84) over the HIP library.

The reference materialises every Stouffer window value in a packed triangle and
searches it recursively.  Here the triangle is implicit: the object keeps the
region's z vector, `getValue` evaluates one window exactly and `segmentTri`
runs the GPU segment search, whose results are identical to the reference's.
"""
import numpy as np

from . import wisetools


class TriArr(object):
    def __init__(self, new_edge):
        self.edge = int(new_edge)
        self.region = np.zeros(self.edge)
        self.ratio = None           # fillTriMin: ratios driving the median-effect filter
        self.mineffectsize = 0
        self.device = 0

    @classmethod
    def from_region(cls, region, device=0, ratio=None, mineffectsize=0):
        tri = cls(len(region))
        tri.region = np.ascontiguousarray(region, dtype=np.float64)
        tri.device = device
        if mineffectsize != 0:
            tri.ratio = np.ascontiguousarray(ratio, dtype=np.float64)
            tri.mineffectsize = mineffectsize
        return tri

    def _ratios(self, start, end):
        return None if self.ratio is None else [self.ratio[start:end]]

    def getValue(self, x, y):
        """Window value as fillTri / fillTriMin stored it (triarray.py:28-29, wisetools.py:471, 483-486)."""
        whole, _ = wisetools.stouffer_segments([self.region[x:y + 1]], np.inf, device=self.device,
                                               ratios=self._ratios(x, y + 1), mineffectsize=self.mineffectsize)
        return whole[0]

    def getSubTriangle(self, start, end):
        """Windows inside [start, end) (triarray.py:31-38)."""
        return TriArr.from_region(self.region[start:end], device=self.device,
                                  ratio=None if self.ratio is None else self.ratio[start:end],
                                  mineffectsize=self.mineffectsize)

    def linTo2D(self, pos):
        """Packed position -> (x, y) (triarray.py:46-51); pure index arithmetic."""
        cur = self.edge
        while pos >= cur:
            pos -= cur
            cur -= 1
        return self.edge - cur, pos + self.edge - cur

    def segmentTri(self, threshold, min_search=3):
        """Recursive most-significant-segment calling (triarray.py:59-84)."""
        _, segs = wisetools.stouffer_segments([self.region], threshold, min_search, device=self.device,
                                              ratios=self._ratios(0, self.edge), mineffectsize=self.mineffectsize)
        return segs[0]
