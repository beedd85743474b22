"""bert4rec_amd: MI355X-native (gfx950) BERT4Rec training + evaluation hot path behind the factory surface of
maneymarkus/BERT4Rec (dataloaders.get_dataloader_factory, models.BERT4RecModel, trainers.get, evaluation.get)."""
__version__ = "0.1.0"
