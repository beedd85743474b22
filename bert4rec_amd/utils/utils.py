"""Path helpers and the JSON config loader (mirrors bert4rec/utils/utils.py:10-47).  Unlike the reference, importing this
module does not raise when VIRTUAL_ENV is unset (utils.py:10 is an import-time KeyError there)."""
import json
import os
import pathlib

_DEFAULT_MODEL_SAVE_PATH = pathlib.Path("saved_models")


def get_virtual_env_path() -> pathlib.Path:
    env = os.environ.get("VIRTUAL_ENV")
    if env is None:
        raise KeyError("VIRTUAL_ENV is not set")
    return pathlib.Path(env)


def get_project_root() -> pathlib.Path:
    return pathlib.Path(__file__).resolve().parent.parent.parent


def get_default_model_save_path() -> pathlib.Path:
    return _DEFAULT_MODEL_SAVE_PATH


def load_json_config(path) -> dict:
    """bert4rec/utils/utils.py:26-40"""
    path = pathlib.Path(path)
    if not path.is_file():
        raise ValueError(f"The given path {path} does not point to a file.")
    with open(path, "r") as f:
        return json.load(f)
