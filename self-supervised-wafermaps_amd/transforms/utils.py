# Normalisation statistics of the wafer-map datasets (reference: src/ssl_wafermap/transforms/utils.py:1-4).
NORMALIZE_STATS = {
    "mean": [0.4496, 0.4496, 0.4496],
    "std": [0.2926, 0.2926, 0.2926],
}
