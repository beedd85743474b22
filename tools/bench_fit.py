"""steps per second through the reference-shaped surface (trainers.get(...).train -> BERT4RecModel.fit) on synthetic
ML-1M-shaped batches, to compare with bench.py's bare train steps"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bert4rec_amd import config, dataloaders, models, trainers
from bert4rec_amd.models.components import networks
from synth import synthetic_batch

V, B, L, P, NB = 3709, 256, 200, 40, 64
enc = networks.Bert4RecEncoder(V, **config.get_encoder_config("ml-1m_64"))
model = models.BERT4RecModel(enc)
trainer = trainers.get(model=model)
trainer.initialize_model()
batches = [synthetic_batch(B, L, P, V, seed=i, rate=0.2) for i in range(NB)]
ds = dataloaders.dataloader_utils.BatchedDataset(batches) if hasattr(dataloaders.dataloader_utils, "BatchedDataset") else batches
t0 = time.perf_counter()
hist = trainer.train(ds, None, epochs=1)
torch.cuda.synchronize(); t1 = time.perf_counter()
hist = trainer.train(ds, None, epochs=3)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"first epoch (incl. upload): {(t1 - t0) / NB * 1e3:.3f} ms/step; next 3 epochs: {(t2 - t1) / (3 * NB) * 1e3:.3f} ms/step "
      f"= {3 * NB * B * P / (t2 - t1) / 1e6:.2f} M masked positions/s; last loss {hist.history['loss'][-1]:.4f}")
