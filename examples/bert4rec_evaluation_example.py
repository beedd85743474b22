"""Evaluate a saved model (the reference's examples/bert4rec_evaluation_example.py): load wrapper + tokenizer, rebuild the test
split with that vocabulary, rank the ground truth among 100 sampled negatives per user on the GPU, print HR / NDCG / MAP.
Trains a small model first when no saved model is given on the command line."""
import pathlib
import sys

from _common import run  # also puts the repository root on sys.path

from bert4rec_amd import dataloaders, datasets, evaluation, models  # noqa: E402
from bert4rec_amd.dataloaders import dataloader_utils  # noqa: E402


def main(path=None):
    synthetic = dict(n_users=1500, n_items=3000, min_len=20, max_len=250, seed=0, order=0.6)
    if path is None:
        run("ml_1m", "ml-1m_64", epochs=3, save_as="saved_models/bert4rec_eval_example", synthetic=synthetic)
        path = "saved_models/bert4rec_eval_example"
    loaded = models.BERT4RecModelWrapper.load(pathlib.Path(path), mode=2)
    model, tokenizer = loaded["model_wrapper"].model, loaded["tokenizer"]
    kwargs = {"tokenizer": tokenizer}
    if not datasets.ML1M.is_available():
        kwargs["data_source"] = datasets.synthetic_dataset(**synthetic)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**kwargs)
    _, _, test_ds = dataloader.prepare_training(device_masking=True)
    for sampler in ("pop_random", "random"):
        evaluator = evaluation.get(dataloader=dataloader, sampler=sampler)
        evaluator.evaluate(model, dataloader_utils.make_batches(test_ds, batch_size=256))
        print(sampler, evaluator.get_metrics_results())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
