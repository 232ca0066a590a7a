"""Row-band sharding of a picture's 64x64 blocks across ranks, and the layout of the compact per-rank ME result
buffer that is all-gathered (RCCL on GPUs, gloo in the CPU tests).

Open-loop ME has no dependency between 64x64 blocks (reference: Docs/Appendix-Open-Loop-Motion-Estimation.md;
SURVEY §8e), so every rank processes a contiguous band of b64 rows of the same picture against fully replicated
reference pyramids, and the only exchange is the all-gather of per-b64 results.  The rows that do not divide evenly rotate
over the ranks from picture to picture (`rotation`), which balances the ranks over the pictures of one exchange.
"""
import numpy as np

from . import abi

# what the host's entropy coder / MD consumes: the MeSbResults arrays and the per-b64 scalars (~1 KB per b64).  The
# search-level arrays (p_sb_best_sad / mv, 5.4 KB per b64, mostly padding for unused reference slots) stay on the rank.
GATHER_FIELDS = tuple(n for n, _, _ in abi.RESULT_FIELDS if n not in ("hme_sc", "hme_sad", "do_ref", "sb_best_sad", "sb_best_mv"))


def band(h64, rank, world, rotate=0):
    """[row0, row1) of b64 rows owned by `rank`: contiguous bands in rank order, h64 // world rows each, the h64 % world
    left-over rows going one each to the ranks rotate, rotate + 1, ... (mod world)."""
    base, extra = divmod(h64, world)
    size = lambda r: base + (1 if (r - rotate) % world < extra else 0)
    row0 = sum(size(r) for r in range(rank))
    return row0, row0 + size(rank)


def rotation(picture, h64, world):
    """Band rotation of the `picture`-th picture of an exchange: the left-over rows move on from picture to picture, so that
    over the pictures in flight every rank owns the same number of rows (34 rows on 8 ranks: 5,5,4,4,4,4,4,4 for one picture,
    68 rows per rank over 16 pictures)."""
    return (picture * (h64 % world)) % world


def rows_max(h64, world):
    return -(-h64 // world)


class BandLayout:
    """Byte layout of one rank's compact result buffer for `n_pictures` pictures: [picture][field][the rank's rows of that picture * w64][elems]
    -- live rows only.  Ranks may own different numbers of rows (rank_bytes); with the rotation the totals are equal whenever
    n_pictures * (h64 % world) is a multiple of world (16 pictures of 34 rows on 8, 4 or 2 ranks).  `nbytes` is the size every rank
    allocates (the largest rank's bytes), `offsets()` / `rank_bytes` are what an all-gather-v needs."""

    def __init__(self, w64, h64, world, n_pu, max_refs, max_cand, n_pictures=1):
        self.w64, self.h64, self.world, self.n_pictures = w64, h64, world, n_pictures
        self.fields = [(n, np.dtype(dt), c(n_pu, max_refs, max_cand)) for n, dt, c in abi.RESULT_FIELDS if n in GATHER_FIELDS]
        self.bytes_per_b64 = sum(dt.itemsize * c for _, dt, c in self.fields)
        self.bands = [[band(h64, r, world, rotation(p, h64, world)) for p in range(n_pictures)] for r in range(world)]
        self.rank_rows = [sum(r1 - r0 for r0, r1 in self.bands[r]) for r in range(world)]
        # every field's band starts on a 16-byte boundary of the rank's buffer (the kernel stores 32-bit fields with dword stores:
        # a band of u8 entries whose length is not a multiple of 4 must not push the next field off its alignment)
        self._foff = [[self._layout_picture(r, p) for p in range(n_pictures)] for r in range(world)]
        off = [0] * world
        for r in range(world):
            for p in range(n_pictures):
                fo, size = self._foff[r][p]
                self._foff[r][p] = {n: (o + off[r], per) for n, (o, per) in fo.items()}
                off[r] += size
        self.rank_bytes = off
        self.nbytes = max(self.rank_bytes)
        self.uniform = len(set(self.rank_bytes)) == 1  # a plain all-gather moves live bytes only

    ALIGN = 16

    def _layout_picture(self, rank, picture):
        r0, r1 = self.bands[rank][picture]
        nb = (r1 - r0) * self.w64
        off, out = 0, {}
        for n, dt, c in self.fields:
            out[n] = (off, dt.itemsize * c)
            off += -(-(nb * dt.itemsize * c) // self.ALIGN) * self.ALIGN
        return out, off

    def band(self, picture, rank):
        """[row0, row1) of `rank` in the `picture`-th picture of the exchange (may be empty when world > h64)."""
        return self.bands[rank][picture]

    def offsets(self):
        """Byte offset of every rank's buffer in the gathered buffer: all-gather layout (rank r at r * nbytes)."""
        return [r * self.nbytes for r in range(self.world)]

    def field_offsets(self, picture, rank):
        """name -> (byte offset of the field's first band row in `rank`'s buffer, bytes per b64)."""
        return self._foff[rank][picture]

    def results_struct(self, base_ptr, picture, rank):
        """abi.MeResults whose pointers are biased so that absolute b64 indices of `rank`'s band land in its compact buffer."""
        first = self.band(picture, rank)[0] * self.w64
        res = abi.MeResults()
        for n, (off, per) in self.field_offsets(picture, rank).items():
            setattr(res, n, base_ptr + off - first * per)
        return res

    def unpack(self, gathered, picture):
        """gathered: uint8 array [world, nbytes] (all ranks' compact buffers).  Returns name -> full-picture array [n_b64, elems]."""
        gathered = np.asarray(gathered).reshape(self.world, self.nbytes)
        out = {n: np.zeros((self.w64 * self.h64, c), dt) for n, dt, c in self.fields}
        for r in range(self.world):
            r0, r1 = self.band(picture, r)
            nb = (r1 - r0) * self.w64
            fo = self.field_offsets(picture, r)
            for n, dt, c in self.fields:
                off, per = fo[n]
                out[n][r0 * self.w64:r1 * self.w64] = gathered[r, off:off + nb * per].view(dt).reshape(nb, c)
        return out
