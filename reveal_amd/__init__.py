"""reveal_amd -- MI355X-native (gfx950, HIP) implementation of REVEAL's recursive
exact-matching hot path (`reveal rem` / reveallib): suffix array + LCP
construction, MUM / multi-MUM scan and the recursive SA/LCP interval split,
behind the reference's own `reveallib.index` API.

    from reveal_amd import reveallib          # 32-bit suffix arrays
    from reveal_amd import reveallib64        # 64-bit suffix arrays (`--64`)
"""
__all__ = ["reveallib", "reveallib64", "rem"]
