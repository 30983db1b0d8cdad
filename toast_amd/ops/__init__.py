"""Operator surface of the map-making hot path, with the reference's class names
(``toast.ops.*``): drop-in for that path."""

from .arithmetic import Combine
from .mapmaker import ApplyAmplitudes, MapMaker, SolveAmplitudes
from .mapmaker_ops import (
    BinMap,
    BuildHitMap,
    BuildInverseCovariance,
    BuildNoiseWeighted,
    Copy,
    CovarianceAndHits,
    Delete,
    NoiseWeight,
    ScanMap,
    ScanMask,
)
from .mapmaker_solve import SolverLHS, SolverRHS, TemplateMatrix, solve
from .ground_filter import GroundFilter
from .noise_filter import NoiseFilter
from .operator import Operator
from .pipeline import Pipeline
from .pointing import BuildPixelDistribution, PixelsHealpix, PointingDetectorSimple, StokesWeights
from .sim_ground import SimGround
