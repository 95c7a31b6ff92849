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

# Vertices that may touch the object in the contact loss of the test-time adaptation
# (DVQ-VAE/utils/loss.py:81-93): five fingertip regions and eight palm vertices, run-length
# encoded as above, in the reference's order (thumb, index, middle, ring, little, palm); 204
# distinct vertices.
_CONTACT_RUNS = (
    ((697, 4), (712, 4), (737, 5), (743, 4), (748, 3), (753, 16)),
    ((46, 4), (164, 4), (194, 2), (223, 1), (237, 2), (280, 2), (298, 1), (301, 1), (317, 1), (320, 1), (323, 11), (340, 16)),
    ((356, 4), (375, 2), (386, 2), (396, 2), (402, 2), (413, 1), (429, 1), (433, 12), (452, 5), (459, 9)),
    ((468, 4), (484, 3), (496, 2), (506, 2), (513, 2), (524, 1), (545, 9), (555, 1), (563, 5), (570, 1), (572, 7)),
    ((580, 4), (600, 3), (614, 2), (624, 2), (630, 2), (641, 1), (663, 6), (670, 1), (672, 1), (680, 5), (686, 10)),
    ((73, 1), (96, 1), (98, 2), (772, 1), (774, 2), (777, 1)),
)
CONTACT_VERTS = [v for runs in _CONTACT_RUNS for first, count in runs for v in range(first, first + count)]
