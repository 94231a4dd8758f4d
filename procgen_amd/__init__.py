"""procgen_amd: MI355X-native vectorized Procgen stepper behind the gym3 libenv C ABI."""
import os as _os

# read by the ROCm runtime when it initialises: one hardware queue per game of a joint handle
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .env import ProcgenGym3Env, ENV_NAMES  # noqa: E402,F401
