"""The whole life cycle in one script (the reference's examples/bert4rec_lifecycle_example.py): dataloader -> vocabulary ->
train / validation / test split -> encoder from a named config -> trainer with a custom optimizer and early stopping -> best
checkpoint -> evaluation -> saved results -> wrapper.save -> load it back and check that the reloaded model ranks the same."""
import pathlib

import torch
from _common import config, dataloader_utils, dataloaders, datasets, evaluation, models, networks, trainers

from bert4rec_amd.models import model_utils
from bert4rec_amd.trainers import optimizers

EPOCHS = 3

if __name__ == "__main__":
    kwargs = {"input_duplication_factor": 1}
    if not datasets.ML1M.is_available():
        kwargs["data_source"] = datasets.synthetic_dataset(n_users=1500, n_items=3706, min_len=20, max_len=200, seed=0, order=0.6)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(**kwargs)
    dataloader.generate_vocab()
    train_ds, val_ds, test_ds = dataloader.prepare_training(finetuning_split=0.15, device_masking=True)
    tokenizer = dataloader.get_tokenizer()

    encoder = networks.Bert4RecEncoder(tokenizer.get_vocab_size(), **config.get_encoder_config("ml-1m_64"))
    model = models.BERT4RecModel(encoder)
    wrapper = models.BERT4RecModelWrapper(model)

    trainer = trainers.get(model=model)
    trainer.initialize_model(optimizer=optimizers.get("adamw", init_lr=1e-3, num_train_steps=2000, num_warmup_steps=50))
    trainer.append_callback(trainers.EarlyStopping(monitor="val_loss", patience=10))

    save_path = model_utils.determine_model_path(pathlib.Path("bert4rec_ml-1m_lifecycle"))
    train_b = dataloader_utils.make_batches(train_ds, batch_size=256, remask_each_epoch=True)
    val_b = dataloader_utils.make_batches(val_ds, batch_size=256)
    test_b = dataloader_utils.make_batches(test_ds, batch_size=256)
    trainer.train(train_b, val_b, checkpoint_path=save_path.joinpath("checkpoints"), epochs=EPOCHS)

    evaluator = evaluation.get(dataloader=dataloader)
    evaluator.evaluate(model, test_b)
    evaluator.save_results(save_path)
    print("test metrics:", evaluator.get_metrics_results())

    trainer.update_wrapper_meta_info(wrapper, dataloader)
    wrapper.save(save_path, tokenizer)
    loaded = models.BERT4RecModelWrapper.load(save_path)
    again = evaluation.get(dataloader=dataloader, seed=0)
    first = evaluation.get(dataloader=dataloader, seed=0)
    first.evaluate(model, test_b)
    again.evaluate(loaded["model_wrapper"].model, test_b)
    assert first.get_metrics_results() == again.get_metrics_results(), "the reloaded model must rank like the saved one"
    print("reloaded model: identical metrics")
