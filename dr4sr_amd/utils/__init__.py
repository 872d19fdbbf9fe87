from .arguments import get_default_parser
from .config import load_config, setup_environment, seed_everything, set_device, get_model_class, prepare_datasets, prepare_model
from .logger import get_logger
