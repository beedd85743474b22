"""Train + evaluate BERT4Rec on ML-1M: the factory call sequence of the reference's examples/bert4rec_ml_1m_example.py:14-91
(without its TensorFlow import).  Reads ml-1m/ratings.dat + movies.dat from $B4R_DATA_DIR/ml-1m (or ./datasets/ml-1m);
falls back to a synthetic Zipf interaction log when the files are absent (there is no network on the GPU boxes)."""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

from bert4rec_amd import config, dataloaders, datasets, evaluation, models, trainers  # noqa: E402
from bert4rec_amd.dataloaders import dataloader_utils  # noqa: E402
from bert4rec_amd.models.components import networks  # noqa: E402


def main(epochs: int = 3, batch_size: int = 256, duplication: int = 5, config_name: str = "ml-1m_64"):
    kwargs = {"input_duplication_factor": duplication}
    if not datasets.ML1M.is_available():
        print("ml-1m not found locally: using a synthetic Zipf log with the same columns")
        kwargs["data_source"] = datasets.synthetic_dataset(n_users=2000, n_items=3000, min_len=20, max_len=300, seed=0)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**kwargs)
    train_ds, val_ds, test_ds = dataloader.prepare_training()

    encoder_config = config.get_encoder_config(config_name)
    encoder = networks.Bert4RecEncoder(dataloader.get_tokenizer().get_vocab_size(), **encoder_config)
    model = models.BERT4RecModel(encoder)

    trainer = trainers.get(model=model)
    trainer.initialize_model()
    trainer.append_callback(trainers.EarlyStopping(monitor="val_loss", patience=2))

    train_batches = dataloader_utils.make_batches(train_ds, batch_size=batch_size)
    val_batches = dataloader_utils.make_batches(val_ds, batch_size=batch_size)
    trainer.train(train_batches, val_batches, checkpoint_path=pathlib.Path("checkpoints/ml-1m/best"), epochs=epochs)

    evaluator = evaluation.get(dataloader=dataloader)
    test_batches = dataloader_utils.make_batches(test_ds, batch_size=batch_size)
    evaluator.evaluate(model, test_batches)
    print(evaluator.get_metrics_results())

    wrapper = models.BERT4RecModelWrapper(model)
    trainer.update_wrapper_meta_info(wrapper, dataloader)
    wrapper.save(pathlib.Path("saved_models/bert4rec_ml1m"), dataloader.get_tokenizer(), mode=2)


if __name__ == "__main__":
    main()
