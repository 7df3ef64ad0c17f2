"""universal_differential_equations_amd -- MI355X-native core for the UDE hot path of
ChrisRackauckas/universal_differential_equations: NN-augmented ODE right-hand sides inside fused adaptive
Tsit5/Vern7 kernels with an interpolating-adjoint backward kernel."""
from . import models  # noqa: F401
from ._lib import UdeError  # noqa: F401
from .sciml import (DeviceEnsemble, Engine, EnsembleMI355, EnsembleProblem, FastInterpolatingAdjoint, ForwardDiffSensitivity,  # noqa: F401
                    InterpolatingAdjoint,
                    ODEProblem, ReverseDiffVJP, Tsit5, Vern7, adjoint_pullback, concrete_solve,
                    loss_and_gradient, remake, rhs, solve)
