"""Constants of the data layout (same values and meaning as /root/reference/jqc/constants.py:21-38)."""

LMAX = 4            # maximum angular momentum (g)
NPRIM_MAX = 3       # primitives per split shell
BASIS_STRIDE = 12   # [x, y, z, ao_loc | c0, e0, c1, e1, c2, e2 | nprim, l]
TILE = 4            # group alignment asked for by apply() for the JK layout (kept for API parity)

# slots 10/11 of a packed row are unused by the reference; this build stores nprim and l there
SLOT_NPRIM = 10
SLOT_ANG = 11



def tile_width(l: int) -> int:
    """Shells per tile edge of the tiled J/K kernels (joltqc_amd/csrc/kernels/jk_tile.hip: ts_of)."""
    return 4 if l <= 2 else (2 if l == 3 else 1)


__all__ = ["LMAX", "NPRIM_MAX", "BASIS_STRIDE", "TILE", "SLOT_NPRIM", "SLOT_ANG", "tile_width"]
