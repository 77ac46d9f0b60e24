"""mpc_amd -- MI355X-native batched QP solve for linear MPC (the hot path of AOE-khkhan/MPC's
src/Planner scripts).  Product code only: HIP kernels + C-ABI in csrc/, ctypes loader, Planner facade."""
from .planner import Planner  # noqa: F401
