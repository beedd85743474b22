"""Saving and loading (the reference's examples/bert4rec_save_and_load_example.py): the wrapper writes the weights under the
reference's Keras variable names (safetensors), the encoder config, the meta config and the tokenizer's vocabulary; load()
returns the wrapper and the tokenizer.  The check at the end: same inputs, same logits."""
import pathlib

import torch
from _common import config, dataloader_utils, dataloaders, datasets, models, networks, trainers

if __name__ == "__main__":
    source = datasets.synthetic_dataset(n_users=300, n_items=500, min_len=10, max_len=60, seed=1)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=source)
    dataloader.generate_vocab()
    tokenizer = dataloader.get_tokenizer()
    model = models.BERT4RecModel(networks.Bert4RecEncoder(tokenizer.get_vocab_size(), **config.get_encoder_config("ml-1m_64")))
    wrapper = models.BERT4RecModelWrapper(model)
    # like Keras, the wrapper refuses to save a model that was never compiled and stepped (bert4rec_wrapper.py:63-68)
    trainer = trainers.get(model=model)
    trainer.initialize_model()
    train, val, _ = dataloader.prepare_training(device_masking=True)
    trainer.train(dataloader_utils.make_batches(train, batch_size=64), dataloader_utils.make_batches(val, batch_size=64), epochs=1)
    path = pathlib.Path("saved_models/bert4rec_save_and_load")
    wrapper.save(path, tokenizer)

    loaded = models.BERT4RecModelWrapper.load(path)
    model2, tokenizer2 = loaded["model_wrapper"].model, loaded["tokenizer"]
    assert tokenizer2.get_vocab_size() == tokenizer.get_vocab_size()
    batch = dataloader.prepare_inference(tokenizer.detokenize([5, 17, 3]))
    a, b = model(batch)["mlm_logits"], model2(batch)["mlm_logits"]
    assert torch.equal(a, b)
    print("saved to", path, "- reloaded model gives bit-identical logits")
