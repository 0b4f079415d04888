"""Skeleton topology constants (oracle copy; test infrastructure only).

Values restate reference Config/config.py:22-55.  Joint ids are indices into
the 21-joint skeleton (0 = pelvis, 20 = head).
"""

JOINTS_ALL = 21
JOINTS_UPPER = 15
JOINTS_LOWER = 8
LOWER_POINTS = 64          # Config.lower_pc_no, config.py:21

# (parent, child) bones, walk order matters for FK (config.py:36-43)
BONES_UPPER = ((20, 3), (3, 2), (2, 1), (2, 4), (2, 8), (4, 5), (5, 6), (6, 7),
               (8, 9), (9, 10), (10, 11), (1, 0), (0, 12), (0, 16))
BONES_LOWER = ((12, 13), (13, 14), (14, 15), (16, 17), (17, 18), (18, 19))
BONES_ALL = BONES_UPPER + BONES_LOWER

UPPER_MAP = (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20)   # config.py:51
LOWER_MAP = (12, 13, 14, 15, 16, 17, 18, 19)                      # config.py:53
LOWER_ROT_MAP = (13, 14, 15, 17, 18, 19)                          # Lower_Net.py:29

# 15-node graph over UPPER_MAP slots used by the ST-GCN (config.py:45-47)
GCN_EDGES = ((0, 12), (0, 13), (0, 1), (1, 2), (2, 3), (2, 4), (2, 8), (3, 14),
             (4, 5), (5, 6), (6, 7), (8, 9), (9, 10), (10, 11))


def upper_fk_plan():
    """[(parent_slot, child_slot, body_row)] for the 14 upper bones.

    Reference Upper_Net.py:138-142: bone i=(parent, child) writes output slot
    UPPER_MAP.index(child) from slot UPPER_MAP.index(parent), rotating
    body[:, i] by q[:, UPPER_MAP.index(child)].
    """
    return [(UPPER_MAP.index(p), UPPER_MAP.index(c), i) for i, (p, c) in enumerate(BONES_UPPER)]


def lower_fk_plan():
    """[(parent_slot, child_slot, rot_row, body_row)] for the 6 leg bones.

    Reference Lower_Net.py:29-35: body row is i+14, rotation row is
    LOWER_ROT_MAP.index(child); slots 0 and 4 are seeded with the hips.
    """
    return [(LOWER_MAP.index(p), LOWER_MAP.index(c), LOWER_ROT_MAP.index(c), i + 14)
            for i, (p, c) in enumerate(BONES_LOWER)]
