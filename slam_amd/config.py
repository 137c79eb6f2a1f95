"""Hot-path configuration: the module-level constants of the reference gathered in one frozen record.

Defaults are the reference's: utils/utills.py:23-24 (RATIO, MATCHES_NORM), :34 (MATCH_METHOD), :39-41
(PNP_NUM_PTS, SUPP_ERR, REC_ERR), :490-491 (prob, initial outlier share, iteration cap)."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .synth import K_KITTI, M1_KITTI, M2_KITTI

NORM_HAMMING = 6
NORM_HAMMING2 = 7
BRUTE_FORCE = "BRUTE FORCE"
REGULAR_KNN = "REGULAR KNN"
FLANN_KNN = "FLANN KNN"
# cv2.BFMatcher(crossCheck=True) has had two behaviours and the reference pins no OpenCV version:
OCV_LEGACY = "ocv_legacy"    # batchDistance(crosscheck=true) up to OpenCV 4.5: train rows vote, queries keep the best voter
MUTUAL = "mutual"            # newer OpenCV: a pair is kept only if each row is the other's nearest


@dataclass(frozen=True)
class HotPathConfig:
    K: np.ndarray = field(default_factory=lambda: K_KITTI.copy())
    M1: np.ndarray = field(default_factory=lambda: M1_KITTI.copy())
    M2: np.ndarray = field(default_factory=lambda: M2_KITTI.copy())
    norm: int = NORM_HAMMING2
    match_method: str = BRUTE_FORCE  # BRUTE_FORCE (cross-check) or REGULAR_KNN (two nearest + ratio test), utills.py:105-119
    ratio: float = 0.6               # RATIO, used by REGULAR_KNN only
    crosscheck_rule: str = OCV_LEGACY
    rec_err: float = 2.0
    supp_err: float = 2.0            # compared as distance <= supp_err**2 (utills.py:715-716, :734-735)
    pnp_num_pts: int = 4
    prob: float = 0.9999
    max_iterations: float = 1e3
    fixed_iterations: int = 0        # > 0 disables the adaptive stop (stress configuration)
    max_draws: int = 0               # 0 = 64 x max_iterations; bounds the reference's unbounded retry loop
    first_chunk: int = 64            # hypotheses evaluated speculatively before the first host replay

    def match_mode(self) -> int:
        """The SLAM_MATCH_* mode of the C ABI this configuration selects."""
        if self.match_method == BRUTE_FORCE:
            if self.crosscheck_rule == OCV_LEGACY:
                return 0
            if self.crosscheck_rule == MUTUAL:
                return 2
            raise ValueError(f"unknown cross-check rule {self.crosscheck_rule!r}")
        if self.match_method == REGULAR_KNN:
            return 1
        raise NotImplementedError(f"match method {self.match_method!r}: FLANN-LSH matching is approximate and "
                                  "non-deterministic (out of scope); use BRUTE_FORCE or REGULAR_KNN")
