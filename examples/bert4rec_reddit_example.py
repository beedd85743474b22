"""BERT4Rec on the Reddit log (vocab 335 420 + 3: the widest item table of the shipped configurations; whole-vocabulary ranking
takes the radix-argsort path of b4r_rank_candidates).  The reference's examples/bert4rec_reddit_example.py flow."""
from _common import run

if __name__ == "__main__":
    run("reddit", "reddit_128", epochs=3, save_as="saved_models/bert4rec_reddit",
        synthetic=dict(n_users=3000, n_items=335420, min_len=5, max_len=60, seed=0, order=0.5))
