"""The masked loss and the two accuracies by hand (the reference's examples/loss_calculation_example.py): the batch-global mean
sum(ce * (y != 0)) / sum(y != 0) of trainer_utils.py:12-23 -- one fused kernel pass over the logits here."""
import torch
from _common import config, dataloader_utils, dataloaders, datasets, models, networks

from bert4rec_amd.trainers import trainer_utils

if __name__ == "__main__":
    source = datasets.synthetic_dataset(n_users=200, n_items=300, min_len=5, max_len=40, seed=3)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=source)
    dataloader.generate_vocab()
    train, _, _ = dataloader.prepare_training()
    batch = next(iter(dataloader_utils.make_batches(train, batch_size=16)))
    model = models.BERT4RecModel(networks.Bert4RecEncoder(dataloader.get_tokenizer().get_vocab_size(),
                                                          **config.get_encoder_config("ml-1m_64")))
    logits = model(batch)["mlm_logits"]
    y = torch.as_tensor(batch["masked_lm_ids"])
    loss = trainer_utils.MaskedSparseCategoricalCrossentropy()(y, logits)
    print("loss", float(loss), " masked accuracy", float(trainer_utils.masked_accuracy(y, logits)),
          " accuracy over all slots", float(trainer_utils.sparse_categorical_accuracy(y, logits)))
    # the same number from its definition
    lp = torch.log_softmax(logits.float().cpu(), -1)
    w = (y != 0).float()
    print("by definition", float(-(lp.gather(-1, y.unsqueeze(-1)).squeeze(-1) * w).sum() / w.sum()))
