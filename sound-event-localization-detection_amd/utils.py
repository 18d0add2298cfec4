"""Logging / device helpers (drop-in for the reference's ``utils.py``) plus the small pieces of
process-group plumbing the data-parallel trainer needs."""
import logging
import os
import sys
from datetime import datetime

import numpy as np
import torch

LOGGER_NAME = "SMR_SELD"


def setup_logging(log_dir="logs", experiment_name="smr_seld"):
    """File + stdout logger named 'SMR_SELD' (utils.py:8-42).  Returns (logger, log_file).
    Under torchrun only rank 0 logs at INFO; other ranks log warnings and above."""
    os.makedirs(log_dir, exist_ok=True)
    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    rank = get_rank()
    suffix = "" if rank == 0 else f"_rank{rank}"
    log_file = os.path.join(log_dir, f"{experiment_name}_{stamp}{suffix}.log")
    logger = logging.getLogger(LOGGER_NAME)
    logger.setLevel(logging.INFO if rank == 0 else logging.WARNING)
    logger.handlers.clear()
    fmt = logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    for handler in (logging.FileHandler(log_file), logging.StreamHandler(sys.stdout)):
        handler.setLevel(logging.INFO)
        handler.setFormatter(fmt)
        logger.addHandler(handler)
    return logger, log_file


def get_rank() -> int:
    return int(os.environ.get("RANK", "0"))


def get_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def get_world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def get_device(logger=None):
    """utils.py:44-63.  On PyTorch-ROCm ``torch.cuda.is_available()`` is true, so main.py needs no
    change on MI355X; under torchrun each process gets ``cuda:{LOCAL_RANK}``."""
    if torch.cuda.is_available():
        local = get_local_rank() if get_world_size() > 1 else torch.cuda.current_device()
        torch.cuda.set_device(local)
        device = torch.device("cuda", local) if get_world_size() > 1 else torch.device("cuda")
        if logger:
            props = torch.cuda.get_device_properties(local)
            logger.info(f"GPU available: {torch.cuda.get_device_name(local)} "
                        f"({props.multi_processor_count} CUs, {props.total_memory / 1024**3:.1f} GB)")
            logger.info(f"HIP version: {getattr(torch.version, 'hip', None)}  CUDA-compat: {torch.version.cuda}")
        torch.backends.cudnn.benchmark = True      # MIOpen find mode (utils.py:53-54)
        torch.backends.cudnn.enabled = True
    else:
        device = torch.device("cpu")
        if logger:
            logger.warning("No GPU visible. Using CPU (plumbing mode: stock PyTorch ops, no HIP kernels).")
    return device


def safe_torch_load(path, map_location=None):
    """utils.py:65-75: full (non weights-only) load of checkpoints THIS code wrote (they pickle the
    Config instance, trainer.py:284)."""
    try:
        return torch.load(path, map_location=map_location, weights_only=False)
    except TypeError:
        return torch.load(path, map_location=map_location)


def polar_to_grid(phi, theta, I=None, J=None, cell_size_deg=None):
    """(azimuth, elevation) in degrees -> (i, j) grid indices (utils.py:77-90): float64 normalise,
    clip to the last cell, truncate."""
    if (I is None or J is None) and cell_size_deg is not None:
        I, J = int(180 // cell_size_deg), int(360 // cell_size_deg)
    elif I is None or J is None:
        raise ValueError("Either provide (I, J) or cell_size_deg for polar_to_grid")
    j = int(np.clip((phi + 180.0) / 360.0 * J, 0, J - 1))
    i = int(np.clip((theta + 90.0) / 180.0 * I, 0, I - 1))
    return i, j
