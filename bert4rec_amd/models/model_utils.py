"""mirrors bert4rec/models/model_utils.py:8-38 (path helper)."""
import pathlib

from ..utils import utils


def determine_model_path(path: pathlib.Path, mode: int = 0) -> pathlib.Path:
    path = pathlib.Path(path)
    if path.is_absolute():
        return path
    if mode == 0:
        return utils.get_project_root().joinpath(utils.get_default_model_save_path()).joinpath(path)
    if mode == 1:
        return utils.get_virtual_env_path().joinpath(utils.get_default_model_save_path()).joinpath(path)
    if mode == 2:
        return path
    raise ValueError(f"The mode parameter has to be in the range of [0, 1, 2], but is: {mode}")
