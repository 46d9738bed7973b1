"""The kernel header and the oracle must carry the same contract constants."""
import os
import re

from oracle import constants as K

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "gsplat_amd_constants.h")


def _defines():
    out = {}
    for line in open(HDR):
        m = re.match(r"#define\s+(GSA_\w+)\s+(.+?)\s*(/\*.*)?$", line)
        if m:
            expr = m.group(2).strip().strip("()")
            expr = re.sub(r"(?<=[0-9.])f\b", "", expr)
            out[m.group(1)] = eval(expr)
    return out


def test_header_matches_oracle():
    d = _defines()
    pairs = {
        "GSA_TILE_SIZE": K.TILE_SIZE, "GSA_NEAR_PLANE": K.NEAR_PLANE, "GSA_FAR_PLANE": K.FAR_PLANE,
        "GSA_EPS2D": K.EPS2D, "GSA_RADIUS_CLIP": K.RADIUS_CLIP, "GSA_ALPHA_THRESHOLD": K.ALPHA_THRESHOLD,
        "GSA_MAX_ALPHA": K.MAX_ALPHA, "GSA_TRANSMITTANCE_EPS": K.TRANSMITTANCE_EPS,
        "GSA_RADIUS_EXTEND": K.RADIUS_EXTEND, "GSA_FOV_MARGIN": K.FOV_MARGIN,
        "GSA_DEPTH_ALPHA_EPS": K.DEPTH_ALPHA_EPS, "GSA_SH_C0": K.SH_C0, "GSA_SH_C1": K.SH_C1,
    }
    for i, v in enumerate(K.SH_C2):
        pairs[f"GSA_SH_C2_{i}"] = v
    for i, v in enumerate(K.SH_C3):
        pairs[f"GSA_SH_C3_{i}"] = v
    for name, val in pairs.items():
        assert name in d, name
        assert abs(d[name] - val) <= 1e-12 * max(1.0, abs(val)), (name, d[name], val)
    assert set(d) == set(pairs)
