"""
Cell-slab decomposition of the structured unit cube along z (scheme "row-owned").

Rank r of G owns the node planes [c_r, c_{r+1}) with c_r = floor(r * nz / G) (the last rank also the top plane
nz) and builds a LOCAL box of cell layers [c_r - 1, c_{r+1}) (no extra layer for rank 0): every cell
touching an owned node is local, so owned matrix rows are assembled completely without communication.
The lowest local node plane of ranks > 0 and the highest of ranks < G-1 are GHOST planes: their rows are
empty, their vector entries are refreshed from the owner before every SpMV (one plane per neighbour).
Pure index arithmetic; shared by the HIP path (perphil_amd/distributed.py) and the CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Slab:
    nx: int
    ny: int
    nz: int
    world: int
    rank: int
    z_begin: int      # first local cell layer
    z_count: int      # local cell layers
    ghost_lo: bool
    ghost_hi: bool

    @property
    def plane(self) -> int:
        return (self.nx + 1) * (self.ny + 1)

    @property
    def local_planes(self) -> int:
        return self.z_count + 1

    @property
    def n_local(self) -> int:
        return self.plane * self.local_planes

    @property
    def owned_planes(self) -> range:
        """Global node planes owned by this rank."""
        lo = self.z_begin + (1 if self.ghost_lo else 0)
        hi = self.z_begin + self.local_planes - (1 if self.ghost_hi else 0)
        return range(lo, hi)

    @property
    def owned_local(self) -> slice:
        """Owned entries of a local vector."""
        lo = self.plane if self.ghost_lo else 0
        hi = self.n_local - (self.plane if self.ghost_hi else 0)
        return slice(lo, hi)

    @property
    def owned_global(self) -> slice:
        """Where the owned entries sit in the global (single-GPU) numbering."""
        r = self.owned_planes
        return slice(r.start * self.plane, r.stop * self.plane)

    def boundary_local(self):
        """(local node ids, global node ids) of the local nodes lying on the boundary of the unit cube
        (ghost planes included: their Dirichlet values are needed for the lifting)."""
        px, py = self.nx + 1, self.ny + 1
        pz = self.local_planes
        m = np.zeros((pz, py, px), dtype=bool)
        m[:, 0, :] = m[:, -1, :] = True
        m[:, :, 0] = m[:, :, -1] = True
        if self.z_begin == 0:
            m[0] = True
        if self.z_begin + self.z_count == self.nz:
            m[-1] = True
        loc = np.nonzero(m.ravel())[0].astype(np.int64)
        return loc, loc + self.z_begin * self.plane


def make_slab(nx: int, ny: int, nz: int, world: int, rank: int) -> Slab:
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if nz // world < 2:
        raise ValueError(f"nz = {nz} gives fewer than 2 cell layers per rank on {world} ranks")
    # balanced split, also when nz is not a multiple of the number of ranks: rank r owns layers [r nz / G, (r+1) nz / G)
    c0, c1 = (rank * nz) // world, ((rank + 1) * nz) // world
    glo, ghi = rank > 0, rank < world - 1
    zb = c0 - (1 if glo else 0)
    return Slab(nx, ny, nz, world, rank, zb, c1 - zb, glo, ghi)
