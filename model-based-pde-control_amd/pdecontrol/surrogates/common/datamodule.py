"""Lightning data module feeding TBPTT (mirror of the reference's
``pdecontrol/surrogates/common/datamodule.py:11-117``): the training / validation loaders are rebuilt
every epoch with the curriculum's current extrapolation length K (window = tau + K).

``device_data`` (a device, or True for ``cuda:LOCAL_RANK``; it rides the reference's ``**training_config`` channel,
pdecontrol/mbrl/mbrl.py:575-589): the replay is packed into HBM once per datamodule and the loaders yield device batches
(``DeviceBatchLoader``: same windows, same order) -- 0.05-0.25 ms per batch instead of 4 ms of host collation, which
would otherwise starve a 0.4 ms training step."""
import os
from typing import List

from pdecontrol._compat.lightning import pl
from pdecontrol.surrogates.common.dataset import DeviceBatchLoader, DeviceSubSeqStore, PDEDataLoader, SubSeqDataset
from pdecontrol.surrogates.common.schedulers import FuncScheduler, Scheduler


class PDEDataModule(pl.LightningDataModule):
    def __init__(self, data, train: List[int], val: List[int] = None, test: List[int] = None,
                 bootstrapping: bool = True, stransf=None, curriculum: Scheduler = None, iteration: int = 0,
                 tau: int = 5, stride: int = None, target_length: int = None, shuffle: bool = True,
                 batch_size: int = 128, device_data=None, **kwargs):
        super().__init__()
        if device_data is True:
            device_data = f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}"
        self.device_data, self._store = device_data or None, None
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
        if self.device_data is not None:
            if self._store is None:
                self._store = DeviceSubSeqStore(self.data, self.device_data)      # the whole replay, packed once
            return DeviceBatchLoader(dataset, self._store, self.batch_size)
        return PDEDataLoader(dataset, batch_size=self.batch_size, shuffle=False, num_workers=0,
                             collate_fn=PDEDataLoader.sample_collate)

    def train_dataloader(self):
        return self._loader(self.train, self.tau + self._horizon(), self.stride, self.bootstrapping)

    def val_dataloader(self):
        return self._loader(self.val, self.tau + self._horizon(), self.stride, self.bootstrapping)

    def test_dataloader(self):
        return self._loader(self.test, self.tau + self.target_length, self.tau, False)
