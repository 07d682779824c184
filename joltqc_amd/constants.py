"""Constants of the data layout (same values and meaning as /root/reference/jqc/constants.py:21-38)."""

LMAX = 4            # maximum angular momentum (g)
NPRIM_MAX = 3       # primitives per split shell
BASIS_STRIDE = 12   # [x, y, z, ao_loc | c0, e0, c1, e1, c2, e2 | nprim, l]
TILE = 4            # group alignment asked for by apply() for the JK layout (kept for API parity)

# slots 10/11 of a packed row are unused by the reference; this build stores nprim and l there
SLOT_NPRIM = 10
SLOT_ANG = 11



import os as _os

# shells per tile edge of the tiled J/K kernels for l = 0..4 (joltqc_amd/csrc/kernels/jk_tile.hip: ts_of; handed to the
# library with jqc_set_tile_widths).  s tiles are wider: the s-containing classes have the cheapest quartets, so the
# per-tile-pair staging and Fock flush must be shared by more of them.
TILE_WIDTHS = tuple(int(x) for x in _os.environ.get("JQC_TILE_WIDTHS", "8,4,4,2,1").split(","))


def tile_width(l: int) -> int:
    """Shells per tile edge of the tiled J/K kernels."""
    return TILE_WIDTHS[l]


__all__ = ["LMAX", "NPRIM_MAX", "BASIS_STRIDE", "TILE", "SLOT_NPRIM", "SLOT_ANG", "tile_width"]
