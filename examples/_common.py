"""Shared driver of the dataset examples: the factory call sequence of the reference's examples/bert4rec_<dataset>_example.py
(dataloader factory -> prepare_training -> encoder from a named config -> model -> trainer -> evaluator -> wrapper.save), with the
options this implementation adds: batches masked on the GPU (new masks every epoch) and a synthetic log when the dataset files
are not on the machine (the GPU boxes have no network)."""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

from bert4rec_amd import config, dataloaders, datasets, evaluation, models, trainers  # noqa: E402
from bert4rec_amd.dataloaders import dataloader_utils  # noqa: E402
from bert4rec_amd.models.components import networks  # noqa: E402


def run(dataset: str, config_name: str, epochs: int, batch_size: int = 256, duplication: int = 1, sampler: str = "pop_random",
        synthetic: dict = None, save_as: str = None):
    factory = dataloaders.get_dataloader_factory("bert4rec")
    create = getattr(factory, f"create_{dataset}_dataloader")
    source = {"ml_1m": datasets.ML1M, "ml_20m": datasets.ML20M, "steam": datasets.Steam, "beauty": datasets.Beauty,
              "reddit": datasets.Reddit}[dataset]
    kwargs = {"input_duplication_factor": duplication}
    if not source.is_available():
        print(f"{dataset}: dataset files not found locally -> synthetic Zipf log with the same columns ({synthetic})")
        cols = ("uid", "movie_name", "timestamp") if dataset.startswith("ml_") else ("user_id", "item_id")
        kwargs["data_source"] = datasets.synthetic_dataset(columns=cols, **(synthetic or {}))
    dataloader = create(**kwargs)
    # device_masking: the three splits come back as token matrices; make_batches masks every batch on the GPU (b4r_mask_batch)
    train_ds, val_ds, test_ds = dataloader.prepare_training(device_masking=True)

    encoder = networks.Bert4RecEncoder(dataloader.get_tokenizer().get_vocab_size(), **config.get_encoder_config(config_name))
    model = models.BERT4RecModel(encoder)
    trainer = trainers.get(model=model)
    trainer.initialize_model()
    trainer.append_callback(trainers.EarlyStopping(monitor="val_loss", patience=3))

    train_batches = dataloader_utils.make_batches(train_ds, batch_size=batch_size, remask_each_epoch=True)
    val_batches = dataloader_utils.make_batches(val_ds, batch_size=batch_size)
    trainer.train(train_batches, val_batches, checkpoint_path=pathlib.Path(f"checkpoints/{dataset}/best"), epochs=epochs)

    evaluator = evaluation.get(dataloader=dataloader, sampler=sampler)
    evaluator.evaluate(model, dataloader_utils.make_batches(test_ds, batch_size=batch_size))
    print(evaluator.get_metrics_results())

    if save_as:
        wrapper = models.BERT4RecModelWrapper(model)
        trainer.update_wrapper_meta_info(wrapper, dataloader)
        wrapper.save(pathlib.Path(save_as), dataloader.get_tokenizer(), mode=2)
    return model, dataloader, evaluator
