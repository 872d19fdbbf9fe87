"""quickstart.run(config) of the reference (quickstart/run.py:7-31): logger -> datasets -> model -> fit -> evaluate.
wandb is optional (the reference initialises it with mode='disabled')."""
import datetime
import os

from ..utils import get_logger, prepare_datasets, prepare_model


def run(config: dict):
    stamp = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S-%f")
    log_path = f"{config['model']['model']}/{config['data']['dataset']}/{stamp}.log"
    logger = get_logger(log_path)
    logger.info("PID of this process: {}".format(os.getpid()))
    dataset_list = prepare_datasets(config)
    logger.info(config)
    logger.info(dataset_list[0])
    model = prepare_model(config, dataset_list)
    model.fit()
    return model.evaluate()
