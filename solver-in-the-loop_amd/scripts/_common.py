import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def logger(path=None):
    log = logging.getLogger("sol")
    if not log.handlers:
        log.addHandler(logging.StreamHandler())
    log.setLevel(logging.INFO)
    if path:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        log.addHandler(logging.FileHandler(path))
    return log


def select_gpu(gpu):
    """`--gpu` of the reference scripts sets CUDA_VISIBLE_DEVICES (karman_train.py:49).  Same here (HIP honours
    HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES) when the process is not a rank of a launcher, which owns the device choice,
    and as long as no device has been initialised yet."""
    import os
    import torch
    if "LOCAL_RANK" in os.environ or gpu in (None, "", "-1") or torch.cuda.is_initialized():
        return
    os.environ.setdefault("HIP_VISIBLE_DEVICES", str(gpu))
