"""Config / CLI helpers with the reference's names and semantics (mvector/utils/utils.py:9-84)."""
import numpy as np

from mvector.utils.logger import logger


def _strtobool(v):
    """distutils.util.strtobool (gone from Python >= 3.12; the reference uses it at utils.py:34)."""
    v = str(v).lower()
    if v in ('y', 'yes', 't', 'true', 'on', '1'):
        return 1
    if v in ('n', 'no', 'f', 'false', 'off', '0'):
        return 0
    raise ValueError(f'invalid truth value {v!r}')


def print_arguments(args=None, configs=None, title=None):
    def _dump(d, indent):
        for k, v in sorted(d.items()):
            if isinstance(v, dict):
                logger.info(f'{indent}{k}:')
                _dump(v, indent + '\t')
            else:
                logger.info(f'{indent}{k}: {v}')

    if args:
        logger.info('----------- 额外配置参数 -----------')
        _dump(vars(args), '')
        logger.info('------------------------------------------------')
    if configs:
        logger.info(f'----------- {title if title else "配置文件参数"} -----------')
        _dump(configs, '')
        logger.info('------------------------------------------------')


def add_arguments(argname, type, default, help, argparser, **kwargs):
    argparser.add_argument('--' + argname, default=default, type=_strtobool if type == bool else type,
                           help=help + ' 默认: %(default)s.', **kwargs)


class Dict(dict):
    """dict with attribute access (reference utils.py:42-45)."""
    __setattr__ = dict.__setitem__
    __getattr__ = dict.__getitem__


def dict_to_object(dict_obj):
    if not isinstance(dict_obj, dict):
        return dict_obj
    return Dict({k: dict_to_object(v) for k, v in dict_obj.items()})


def cal_accuracy(y_score, y_true, threshold=0.5):
    """Fraction of trials on the right side of ``threshold`` (labels 0 / 1)."""
    decided = np.asarray(y_score) >= threshold
    return float(np.mean(decided == np.asarray(y_true).astype(bool)))


def cal_accuracy_threshold(y_score, y_true):
    """Best accuracy over the thresholds 0.00, 0.01, ... 0.99 and the (first) threshold reaching it, as the reference's
    scan does; all thresholds at once."""
    scores, labels = np.asarray(y_score), np.asarray(y_true).astype(bool)
    grid = np.arange(100) * 0.01
    acc = ((scores[None, :] >= grid[:, None]) == labels[None, :]).mean(axis=1)
    best = int(np.argmax(acc))  # argmax returns the first maximum, like the strict '>' of the scan
    if acc[best] <= 0:
        return 0, 0
    return acc[best], grid[best]


def cosin_metric(x1, x2):
    a, b = np.asarray(x1), np.asarray(x2)
    return float(a @ b) / float(np.linalg.norm(a) * np.linalg.norm(b))
