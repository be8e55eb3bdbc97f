from afdm.training import set_seed, setup_logging  # noqa: F401
from afdm.imageio_utils import save_images, make_grid  # noqa: F401
