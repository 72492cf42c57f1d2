"""ScoreParams: every constant of the hot path lifted out of the reference's inline magic numbers
(SURVEY.md Appendix A); defaults are the reference values (lg_default_params)."""
from ._lib import LgParams, default_params

ScoreParams = LgParams

__all__ = ["ScoreParams", "default_params"]
