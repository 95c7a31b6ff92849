"""Vertex subsets of the 778-vertex MANO mesh fed to the six hand-part encoders.

Five sets (index/middle/little/ring region lists and the palm) are the literal index
lists of the reference (DVQ-VAE/network/DVQVAE.py:54-87), stored run-length encoded
as (first vertex, count) and expanded in order; repeated vertices in the reference's
lists (207 in the palm, 162 and 163 in part 3) are preserved.  Part 0 is not defined
anywhere in the reference (DVQVAE.py:93 reads an unassigned name); it is reconstructed
as the complement of the other five in range(778) -- 83 vertices around the thumb.
Order: [part0, part1, part2, part3, part4, palm], the order DVQVAE.py:93 zips the
part encoders with.
"""

_RUNS = (
    ((240, 1), (248, 6), (266, 2), (286, 2), (697, 72)),
    ((46, 4), (56, 4), (86, 2), (133, 8), (155, 2), (164, 14), (186, 1), (189, 1), (194, 2), (212, 2), (221, 6),
     (237, 2), (245, 1), (258, 4), (263, 1), (272, 3), (280, 4), (294, 62)),
    ((185, 1), (187, 1), (246, 1), (262, 1), (269, 2), (277, 1), (288, 1), (356, 112)),
    ((160, 3), (162, 2), (163, 1), (290, 4), (468, 112)),
    ((201, 2), (278, 1), (289, 1), (580, 117)),
    ((0, 46), (50, 6), (60, 26), (88, 45), (141, 14), (157, 3), (178, 7), (188, 1), (190, 4), (196, 5), (203, 5),
     (207, 5), (214, 7), (227, 10), (239, 1), (241, 4), (247, 1), (254, 4), (264, 2), (268, 1), (271, 1), (275, 2),
     (279, 1), (284, 2), (769, 9)),
)

HAND_PARTS = [[v for first, count in runs for v in range(first, first + count)] for runs in _RUNS]
PART_SIZES = [len(p) for p in HAND_PARTS]  # [83, 123, 120, 122, 121, 212]
