"""quickstart.run(config) of the reference (quickstart/run.py:7-31): logger -> datasets -> model -> fit -> evaluate.
wandb is optional (the reference initialises it with mode='disabled')."""
import datetime
import os

from ..utils import get_logger, prepare_datasets, prepare_model


def _shared_stamp(device=None) -> str:
    """One log / checkpoint stem for the whole job.  EarlyStopping names the best checkpoint after the log file
    (utils/callbacks.py:18-25 of the reference) and only rank 0 writes it, so under torch.distributed.run every rank must use
    RANK 0's stamp — a per-process microsecond stamp would make ranks > 0 look for a checkpoint that was never written."""
    stamp = datetime.datetime.now().strftime("%Y-%m-%d-%H-%M-%S-%f")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        from ..parallel import init_distributed
        init_distributed(device)
        box = [stamp]
        dist.broadcast_object_list(box, src=0)
        stamp = box[0]
    return stamp


def run(config: dict):
    stamp = _shared_stamp(config["train"].get("device"))
    log_path = f"{config['model']['model']}/{config['data']['dataset']}/{stamp}.log"
    logger = get_logger(log_path)
    logger.info("PID of this process: {}".format(os.getpid()))
    dataset_list = prepare_datasets(config)
    logger.info(config)
    logger.info(dataset_list[0])
    model = prepare_model(config, dataset_list)
    model.fit()
    return model.evaluate()
