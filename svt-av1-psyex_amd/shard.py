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
    """Byte layout of one rank's compact result buffer for `n_pictures` pictures: [picture][field][rows_max * w64][elems]."""

    def __init__(self, w64, h64, world, n_pu, max_refs, max_cand, n_pictures=1):
        self.w64, self.h64, self.world, self.n_pictures = w64, h64, world, n_pictures
        self.fields = [(n, np.dtype(dt), c(n_pu, max_refs, max_cand)) for n, dt, c in abi.RESULT_FIELDS if n in GATHER_FIELDS]
        self.nbb = rows_max(h64, world) * w64
        self.bytes_per_b64 = sum(dt.itemsize * c for _, dt, c in self.fields)
        self.nbytes = n_pictures * self.nbb * self.bytes_per_b64

    def band(self, picture, rank):
        """[row0, row1) of `rank` in the `picture`-th picture of the exchange."""
        return band(self.h64, rank, self.world, rotation(picture, self.h64, self.world))

    def field_offsets(self, picture):
        """name -> (byte offset of the field's first band row, bytes per b64)."""
        off = picture * self.nbb * self.bytes_per_b64
        out = {}
        for n, dt, c in self.fields:
            out[n] = (off, dt.itemsize * c)
            off += self.nbb * dt.itemsize * c
        return out

    def results_struct(self, base_ptr, picture, rank):
        """abi.MeResults whose pointers are biased so that absolute b64 indices of `rank`'s band land in its compact buffer."""
        first = self.band(picture, rank)[0] * self.w64
        res = abi.MeResults()
        for n, (off, per) in self.field_offsets(picture).items():
            setattr(res, n, base_ptr + off - first * per)
        return res

    def unpack(self, gathered, picture):
        """gathered: uint8 array [world, nbytes] (all ranks' compact buffers).  Returns name -> full-picture array [n_b64, elems]."""
        gathered = np.asarray(gathered).reshape(self.world, self.nbytes)
        out = {}
        for n, dt, c in self.fields:
            off, per = self.field_offsets(picture)[n]
            full = np.zeros((self.w64 * self.h64, c), dt)
            for r in range(self.world):
                r0, r1 = self.band(picture, r)
                nb = (r1 - r0) * self.w64
                full[r0 * self.w64:r1 * self.w64] = gathered[r, off:off + nb * per].view(dt).reshape(nb, c)
            out[n] = full
        return out
