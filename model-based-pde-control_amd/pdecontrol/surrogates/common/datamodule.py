"""Lightning data module feeding TBPTT (mirror of the reference's
``pdecontrol/surrogates/common/datamodule.py:11-117``): the training / validation loaders are rebuilt
every epoch with the curriculum's current extrapolation length K (window = tau + K)."""
from typing import List

from pdecontrol._compat.lightning import pl
from pdecontrol.surrogates.common.dataset import PDEDataLoader, SubSeqDataset
from pdecontrol.surrogates.common.schedulers import FuncScheduler, Scheduler


class PDEDataModule(pl.LightningDataModule):
    def __init__(self, data, train: List[int], val: List[int] = None, test: List[int] = None,
                 bootstrapping: bool = True, stransf=None, curriculum: Scheduler = None, iteration: int = 0,
                 tau: int = 5, stride: int = None, target_length: int = None, shuffle: bool = True,
                 batch_size: int = 128, **kwargs):
        super().__init__()
        self.data, self.train, self.val, self.test = data, train, val, test
        self.bootstrapping, self.stransf = bootstrapping, stransf
        self.curriculum = FuncScheduler(steptype="epoch", func=lambda *_: 1) if curriculum is None else curriculum
        self.tau, self.stride, self.target_length = tau, stride, target_length
        self.batch_size, self.shuffle, self.iteration = batch_size, shuffle, iteration

    def _horizon(self):
        trainer = self.trainer
        return int(self.curriculum(self.iteration, trainer.current_epoch, trainer.global_step))

    def _loader(self, subsamples, length, stride, bootstrapping):
        dataset = SubSeqDataset(data=self.data, subsamples=subsamples, length=length, stride=stride,
                                bootstrapping=bootstrapping, stransf=self.stransf)
        return PDEDataLoader(dataset, batch_size=self.batch_size, shuffle=False, num_workers=0,
                             collate_fn=PDEDataLoader.sample_collate)

    def train_dataloader(self):
        return self._loader(self.train, self.tau + self._horizon(), self.stride, self.bootstrapping)

    def val_dataloader(self):
        return self._loader(self.val, self.tau + self._horizon(), self.stride, self.bootstrapping)

    def test_dataloader(self):
        return self._loader(self.test, self.tau + self.target_length, self.tau, False)
