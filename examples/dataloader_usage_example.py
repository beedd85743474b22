"""What a dataloader hands to the model (the reference's examples/dataloader_usage_example.py): the six int64 tensors of a batch
(bert4rec_preprocessor.py:48-116), once built on the host like the reference does and once as a token matrix that is masked on the
GPU batch by batch (b4r_mask_batch)."""
from _common import dataloader_utils, dataloaders, datasets

if __name__ == "__main__":
    source = datasets.synthetic_dataset(n_users=200, n_items=300, min_len=5, max_len=40, seed=2)
    dataloader = dataloaders.get_dataloader_factory("bert4rec").create_ml_1m_dataloader(data_source=source)
    dataloader.generate_vocab()
    train, val, test = dataloader.prepare_training()                       # host path: masked once, like the reference
    batch = next(iter(dataloader_utils.make_batches(train, batch_size=4)))
    for k, v in batch.items():
        print(f"{k:22s} {tuple(v.shape)} {v.dtype}  first row: {v[0].tolist()[:12]} ...")
    dtrain, _, _ = dataloader.prepare_training(device_masking=True)        # device path: [U, L] tokens, masks drawn per batch
    print("device path:", type(dtrain).__name__, "tokens", tuple(dtrain.tokens.shape))
    print("inference input for a history of 3 items:", {k: v.tolist() for k, v in dataloader.prepare_inference(dataloader.get_tokenizer().detokenize([3, 4, 5])).items()})
