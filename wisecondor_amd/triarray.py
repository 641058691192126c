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
        self.device = 0

    @classmethod
    def from_region(cls, region, device=0):
        tri = cls(len(region))
        tri.region = np.ascontiguousarray(region, dtype=np.float64)
        tri.device = device
        return tri

    def getValue(self, x, y):
        """sum(z[x..y]) / sqrt(y-x+1) (triarray.py:28-29 reading what wisetools.py:471 stored)."""
        whole, _ = wisetools.stouffer_segments([self.region[x:y + 1]], np.inf, device=self.device)
        return whole[0]

    def getSubTriangle(self, start, end):
        """Windows inside [start, end) (triarray.py:31-38)."""
        return TriArr.from_region(self.region[start:end], device=self.device)

    def linTo2D(self, pos):
        """Packed position -> (x, y) (triarray.py:46-51); pure index arithmetic."""
        cur = self.edge
        while pos >= cur:
            pos -= cur
            cur -= 1
        return self.edge - cur, pos + self.edge - cur

    def segmentTri(self, threshold, min_search=3):
        """Recursive most-significant-segment calling (triarray.py:59-84)."""
        _, segs = wisetools.stouffer_segments([self.region], threshold, min_search, device=self.device)
        return segs[0]
