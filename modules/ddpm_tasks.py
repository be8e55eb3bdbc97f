from afdm.tasks import ddpm_run, rotation_results, shift_results  # noqa: F401
from modules.ddpm_models import *  # noqa: F401,F403
