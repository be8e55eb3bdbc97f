from afdm.training import set_seed, setup_logging  # noqa: F401
from afdm.imageio_utils import save_images, make_grid  # noqa: F401
from afdm.data import get_data, get_data_MNIST, save_gen_images, save_dataset_MNIST, make_collage  # noqa: F401
