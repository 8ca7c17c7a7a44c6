"""POD layouts shared with the C-ABI (include/restir_rt.h).

Field-for-field the reference's structs (layouts measured in SURVEY.md §8a):
Triangle common/core.hpp:38-43 (60 B), Visibility common/core.hpp:167-172 (16 B),
Reservoir common/reservoir.hpp:5-38 (76 B), Options common/options.hpp:4-22 (48 B),
RayGenerator common/camera.hpp:5-9 (36 B).
"""
import numpy as np

TRIANGLE = np.dtype([("v", "<f4", (3, 3)), ("color", "<f4", 3), ("emissive", "<f4", 3)])
VISIBILITY = np.dtype([("uv", "<f4", 2), ("index", "<i4"), ("pad", "<i4")])
RESERVOIR = np.dtype(
    [
        ("origin_position", "<f4", 3),
        ("origin_normal", "<f4", 3),
        ("hit_position", "<f4", 3),
        ("hit_normal", "<f4", 3),
        ("radiance", "<f4", 3),
        ("visibility", "u1"),
        ("pad", "u1", 3),
        ("w_sum", "<f4"),
        ("ucw", "<f4"),
        ("M", "<i4"),
    ]
)
OPTIONS = np.dtype(
    {
        "names": [
            "accumulate", "max_depth", "sky_color", "ris_sample_count",
            "rejection_heuristics_threshold", "use_temporal_resampling", "use_spatial_resampling",
            "spatial_resampling_sample_count", "spatial_resampling_radius",
            "spatial_resampling_passes", "use_shadowed_target_function", "use_visibility_reuse",
        ],
        "formats": ["u1", "<i4", ("<f4", 3), "<i4", "<f4", "u1", "u1", "<i4", "<f4", "<i4", "u1", "u1"],
        "offsets": [0, 4, 8, 20, 24, 28, 29, 32, 36, 40, 44, 45],
        "itemsize": 48,
    }
)
RAYGEN = np.dtype([("origin", "<f4", 3), ("right", "<f4", 3), ("up", "<f4", 3)])

assert TRIANGLE.itemsize == 60 and VISIBILITY.itemsize == 16 and RESERVOIR.itemsize == 76
assert OPTIONS.itemsize == 48 and RAYGEN.itemsize == 36


def default_options(**kw):
    """Defaults of common/options.hpp:6-22, with keyword overrides."""
    o = np.zeros(1, dtype=OPTIONS)
    o["max_depth"] = 6
    o["ris_sample_count"] = 32
    o["rejection_heuristics_threshold"] = 0.2
    o["spatial_resampling_sample_count"] = 5
    o["spatial_resampling_radius"] = 30.0
    o["spatial_resampling_passes"] = 3
    o["use_visibility_reuse"] = 1
    for k, v in kw.items():
        o[k] = v
    return o


def bench_options(**kw):
    """Benchmark options of SURVEY.md §8(d): defaults + keys `1`,`2` (temporal + spatial reuse)."""
    d = dict(use_temporal_resampling=1, use_spatial_resampling=1)
    d.update(kw)
    return default_options(**d)
