"""Golden-vector case table shared by make_golden.py and the tests."""

# name -> ViT dims (v2 Config field values) + batch + seed
VIT_CASES = {
    # BASELINE C1/C2 architecture (K=1 head: SURVEY 8a row a12)
    "c1":     dict(image=32, patch=4, embed=384, heads=4, layers=6, mlp_ratio=2, classes=1, channels=3, batch=2, seed=11),
    # reference default classes_count=10 on the same trunk, fewer layers
    "c1k10":  dict(image=32, patch=4, embed=384, heads=4, layers=2, mlp_ratio=2, classes=10, channels=3, batch=3, seed=12),
    # reference Config() defaults (E=128)
    "e128":   dict(image=32, patch=4, embed=128, heads=4, layers=6, mlp_ratio=2, classes=10, channels=3, batch=2, seed=13),
    # C4 shape, 2 layers
    "c4":     dict(image=64, patch=8, embed=512, heads=8, layers=2, mlp_ratio=2, classes=1, channels=3, batch=2, seed=14),
    # C5 shape, 1 layer
    "c5":     dict(image=128, patch=16, embed=768, heads=12, layers=1, mlp_ratio=2, classes=1, channels=3, batch=1, seed=15),
}

# v1 generator at its config defaults (src/v1/config.py:45-49,60-66)
GEN_CASES = {
    "g1": dict(batch=2, seed=21),
    "g1b3": dict(batch=3, seed=22),
}

# v1 MultiHeadSelfAttention with L2-distance scores (discriminator regulariser of src/v1/attention.py:54-67,73-103)
V1ATT_CASES = {
    "l2": dict(embed=384, heads=4, head_dim=96, seq=65, batch=2, seed=31, spectral=False),
    "l2spec": dict(embed=384, heads=4, head_dim=96, seq=65, batch=2, seed=32, spectral=True),
    "l2e128": dict(embed=128, heads=4, head_dim=32, seq=17, batch=3, seed=33, spectral=False),
}
