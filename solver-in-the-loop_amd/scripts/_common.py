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
