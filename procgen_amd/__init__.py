"""procgen_amd: MI355X-native vectorized Procgen stepper behind the gym3 libenv C ABI."""
from .env import ProcgenGym3Env, ENV_NAMES  # noqa: F401
