from ..mmcv_lite import get_logger


def get_root_logger(log_file=None, log_level='INFO'):
    """mmdet/utils/logger.py:6-19."""
    import logging
    return get_logger('mmdet', log_file, getattr(logging, log_level) if isinstance(log_level, str) else log_level)
