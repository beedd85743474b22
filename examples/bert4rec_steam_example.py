"""BERT4Rec on Steam (BASELINE.json configs[4]: L = 50, P = 20, mask probability 0.4, hidden 64) with the popularity sampler's 100
negatives per user: the reference's examples/bert4rec_steam_example.py flow.  Reads steam.txt from $B4R_DATA_DIR/steam."""
from _common import run

if __name__ == "__main__":
    run("steam", "steam_64", epochs=5, sampler="popular", save_as="saved_models/bert4rec_steam",
        synthetic=dict(n_users=4000, n_items=13044, min_len=5, max_len=60, seed=0, order=0.5))
