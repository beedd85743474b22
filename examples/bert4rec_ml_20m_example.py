"""BERT4Rec on ML-20M (BASELINE.json configs[3]: L = 200, hidden 256, 8 heads, inner 1024): the reference's
examples/bert4rec_ml_20m_example.py flow.  Reads ml-20m/ratings.csv + movies.csv from $B4R_DATA_DIR/ml-20m."""
from _common import run

if __name__ == "__main__":
    run("ml_20m", "ml-20m_256", epochs=2, save_as="saved_models/bert4rec_ml20m",
        synthetic=dict(n_users=3000, n_items=26729, min_len=20, max_len=300, seed=0, order=0.5))
