"""Singleton logger (reference: util/model_log.py:7-50): stream + optional log file."""
import logging
import os
import threading


class create_log(object):
    _instance_lock = threading.Lock()

    def __new__(cls, *args, **kwargs):
        if not hasattr(create_log, "_instance"):
            with create_log._instance_lock:
                if not hasattr(create_log, "_instance"):
                    create_log._instance = object.__new__(cls)
                    create_log._instance._built = False
        return create_log._instance

    def __init__(self, type=None, experiment_type=None, version=None, log_dir="data/log_data"):
        if self._built:
            return
        self._built = True
        self.logger = logging.getLogger("mtamrecommender_amd")
        self.logger.setLevel(logging.INFO)
        if not self.logger.handlers:
            fmt = logging.Formatter("%(asctime)s %(levelname)s %(message)s")
            sh = logging.StreamHandler()
            sh.setFormatter(fmt)
            self.logger.addHandler(sh)
            if type is not None:
                os.makedirs(log_dir, exist_ok=True)
                fh = logging.FileHandler(os.path.join(
                    log_dir, "%s_%s_%s_log.txt" % (type, experiment_type, version)))
                fh.setFormatter(fmt)
                self.logger.addHandler(fh)
