"""Train + evaluate BERT4Rec on ML-1M: the reference's examples/bert4rec_ml_1m_example.py:14-91 statement for statement, minus its
TensorFlow / absl imports (`tf.keras.callbacks.EarlyStopping` -> `trainers.EarlyStopping`; `model(model.inputs)`, which only
builds Keras variables, has no counterpart: the engine allocates its parameters in the constructor).

Defaults ARE the reference's literals (:21-30): 150 epochs, batch 256, input_duplication_factor 5, finetuning_split 0.1,
`ml-1m_128.json` (hidden 128, 4 heads, inner 512), early-stopping config {val_loss, patience 20} -- which the reference builds and
then leaves un-appended (:71, commented out), so by default it is not appended here either.  Arguments exist for short runs:

    python examples/bert4rec_ml_1m_example.py                       # the reference's run
    python examples/bert4rec_ml_1m_example.py --epochs 3 --config ml-1m_64 --early-stopping --patience 2

Reads ml-1m/ratings.dat + movies.dat from $B4R_DATA_DIR/ml-1m (or ./datasets/ml-1m); falls back to a synthetic Zipf interaction
log when the files are absent (there is no network on the GPU boxes)."""
import argparse
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

from bert4rec_amd import config as configs, datasets, trainers  # noqa: E402
from bert4rec_amd.dataloaders import dataloader_utils, get_dataloader_factory  # noqa: E402
from bert4rec_amd.evaluation import BERT4RecEvaluator  # noqa: E402
from bert4rec_amd.models import BERT4RecModel, BERT4RecModelWrapper, model_utils  # noqa: E402
from bert4rec_amd.models.components import networks  # noqa: E402


def main(epochs: int = 150, batch_size: int = 256, input_duplication_factor: int = 5, finetuning_split: float = 0.1,
         encoder_config: str = "ml-1m_128.json", patience: int = 20, append_early_stopping: bool = False,
         save_name: str = "bert4rec_ml-1m_15"):
    # definition of variables (bert4rec_ml_1m_example.py:21-30)
    EPOCHS = epochs
    early_stopping_config = {
        "monitor": "val_loss",
        "patience": patience,
        "verbose": 1,
    }

    dataloader_factory = get_dataloader_factory("bert4rec")
    dataloader_config = {
        "input_duplication_factor": input_duplication_factor
    }
    if not datasets.ML1M.is_available():
        print("ml-1m not found locally: using a synthetic Zipf log with the same columns")
        dataloader_config["data_source"] = datasets.synthetic_dataset(n_users=2000, n_items=3000, min_len=20, max_len=300, seed=0)
    dataloader = dataloader_factory.create_ml_1m_dataloader(**dataloader_config)
    dataloader.generate_vocab()
    train_ds, val_ds, test_ds = dataloader.prepare_training(finetuning_split=finetuning_split)
    tokenizer = dataloader.get_tokenizer()

    # load a specific config (the values of bert4rec/config/bert4rec_train_configs/<encoder_config>)
    config = configs.get_encoder_config(encoder_config)

    bert_encoder = networks.Bert4RecEncoder(tokenizer.get_vocab_size(), **config)
    model = BERT4RecModel(bert_encoder)
    model_wrapper = BERT4RecModelWrapper(model)

    # set up trainer
    trainer = trainers.get(**{"model": model})
    trainer.initialize_model()

    save_path = model_utils.determine_model_path(pathlib.Path(save_name), mode=2)
    # is needed as this does not create a new folder but rather the base name (or prefix)
    # for the created checkpoint files
    checkpoint_path = save_path.joinpath("checkpoints")

    train_batches = dataloader_utils.make_batches(train_ds, batch_size=batch_size)
    val_batches = dataloader_utils.make_batches(val_ds, batch_size=batch_size)
    test_batches = dataloader_utils.make_batches(test_ds, batch_size=batch_size)

    # set up a training loop callback (built and NOT appended in the reference, :70-71)
    early_stopping_callback = trainers.EarlyStopping(**early_stopping_config)
    if append_early_stopping:
        trainer.append_callback(early_stopping_callback)

    model_wrapper.update_meta({
        "EPOCHS": EPOCHS,
        "input_duplication_factor": input_duplication_factor,
        "finetuning_split": finetuning_split,
        "early_stopping_config": early_stopping_config
    })

    # train the model
    trainer.train(train_batches, val_batches, checkpoint_path=checkpoint_path, epochs=EPOCHS)
    trainer.update_wrapper_meta_info(model_wrapper, dataloader)

    evaluator = BERT4RecEvaluator(dataloader=dataloader)

    metrics_objects = evaluator.evaluate(model, test_batches)   # noqa: F841  (kept: the reference binds it too)
    evaluator.save_results(save_path)
    metrics = evaluator.get_metrics_results()
    print(metrics)

    model_wrapper.save(save_path=save_path, tokenizer=tokenizer, mode=2)
    return metrics


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="defaults = the literals of the reference's examples/bert4rec_ml_1m_example.py")
    ap.add_argument("--epochs", type=int, default=150)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--duplication", type=int, default=5)
    ap.add_argument("--finetuning-split", type=float, default=0.1)
    ap.add_argument("--config", default="ml-1m_128.json")
    ap.add_argument("--patience", type=int, default=20)
    ap.add_argument("--early-stopping", action="store_true", help="append the early-stopping callback (the reference leaves it off)")
    ap.add_argument("--save-name", default="bert4rec_ml-1m_15")
    a = ap.parse_args()
    main(a.epochs, a.batch_size, a.duplication, a.finetuning_split, a.config, a.patience, a.early_stopping, a.save_name)
