"""Run-time switches of the HIP path."""

# Fuse get_graph_feature_sv -> binarized SVBlock -> svpool into one pass over the edges (csrc/edgeblock.hip).
# Off = tier 1: every tensor the reference materialises is materialised (used as the on-device cross-check).
FUSE_EDGE_BLOCKS = True
