"""Encoder hyper-parameter sets of the reference (values of bert4rec/config/bert4rec_train_configs/<name>.json),
kept as a table: (hidden_size, inner_dim, num_attention_heads, num_layers, max_sequence_length, attention_dropout,
output_dropout)."""
import json
import pathlib

_TABLE = {
    "beauty_64": (64, 64, 2, 2, 50, 0.2, 0.5), "beauty_128": (128, 512, 4, 2, 50, 0.2, 0.5),
    "beauty_256": (256, 1024, 8, 2, 50, 0.2, 0.5),
    "ml-1m_64": (64, 256, 2, 2, 200, 0.2, 0.2), "ml-1m_128": (128, 512, 4, 2, 200, 0.2, 0.5),
    "ml-1m_256": (256, 512, 8, 2, 200, 0.2, 0.5),
    "ml-20m_64": (64, 256, 2, 2, 200, 0.1, 0.1), "ml-20m_128": (128, 512, 4, 2, 200, 0.1, 0.1),
    "ml-20m_256": (256, 1024, 8, 2, 200, 0.1, 0.1),
    "reddit_128": (128, 512, 4, 2, 200, 0.1, 0.1),
    "steam_64": (64, 256, 2, 2, 50, 0.1, 0.1), "steam_128": (128, 512, 4, 2, 50, 0.1, 0.1),
    "steam_256": (256, 1024, 8, 2, 50, 0.2, 0.2),
}


def available_configs():
    return sorted(_TABLE)


def get_encoder_config(name: str) -> dict:
    """Kwargs for Bert4RecEncoder(vocab_size, **config), e.g. get_encoder_config("ml-1m_64")."""
    key = name[:-5] if name.endswith(".json") else name
    if key not in _TABLE:
        raise ValueError(f"{name} is not a known encoder config; available: {available_configs()}")
    h, inner, heads, layers, L, ad, od = _TABLE[key]
    return {"attention_dropout": ad, "output_dropout": od, "hidden_size": h, "inner_dim": inner,
            "max_sequence_length": L, "num_attention_heads": heads, "num_layers": layers}


def write_config_files(directory) -> list:
    """Materialise <name>.json files (the layout the reference's examples load with utils.load_json_config)."""
    directory = pathlib.Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    out = []
    for name in available_configs():
        p = directory / f"{name}.json"
        with open(p, "w") as f:
            json.dump(get_encoder_config(name), f, indent=4)
        out.append(p)
    return out
