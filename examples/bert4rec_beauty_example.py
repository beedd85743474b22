"""BERT4Rec on Amazon Beauty (vocab 54 542 + 3, tests/datalaoders_tests/bert4rec_dataloaders_tests.py:237): the reference's
examples/bert4rec_beauty_example.py flow.  Reads the ratings file from $B4R_DATA_DIR/beauty."""
from _common import run

if __name__ == "__main__":
    run("beauty", "beauty_64", epochs=5, save_as="saved_models/bert4rec_beauty",
        synthetic=dict(n_users=3000, n_items=54542, min_len=5, max_len=60, seed=0, order=0.5))
