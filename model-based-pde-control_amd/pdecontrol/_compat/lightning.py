"""pytorch_lightning when importable, otherwise the few names the surrogate code touches.

The shim's LightningModule is an ``nn.Module`` whose ``log`` keeps the latest value per name (as a
tensor, no host sync); ``Trainer`` is a minimal single-device fit loop (manual zero_grad /
backward / step over a dataloader) so that tests and bench.py can drive ``training_step`` and
``configure_optimizers`` the way Lightning does.  pytorch-lightning 1.7.2 is the reference's pin
(poetry.lock:1371) and is absent from the build image.
"""
try:  # pragma: no cover - depends on the environment
    import pytorch_lightning as pl  # type: ignore
    IS_SHIM = False
except ImportError:
    import types

    import torch
    from torch import nn

    class LightningModule(nn.Module):
        automatic_optimization = True

        def __init__(self):
            super().__init__()
            self.logged = {}

        # what a module under manual optimization asks its trainer for (pytorch-lightning 1.7 API)
        def optimizers(self):
            opts = self.__dict__.get("_shim_optimizers")
            if not opts:
                raise RuntimeError("no trainer attached")
            return opts[0] if len(opts) == 1 else opts

        def lr_schedulers(self):
            scheds = self.__dict__.get("_shim_schedulers")
            if not scheds:
                return None
            return scheds[0] if len(scheds) == 1 else scheds

        def on_train_epoch_end(self):
            pass

        def log(self, name, value, *args, **kwargs):
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        @property
        def device(self):
            p = next(self.parameters(), None)
            return p.device if p is not None else torch.device("cpu")

    class LightningDataModule:
        pass

    class Callback:
        pass

    class Trainer:
        def __init__(self, max_steps=-1, max_epochs=1, gradient_clip_val=None, **kwargs):
            self.max_steps, self.max_epochs, self.gradient_clip_val = max_steps, max_epochs, gradient_clip_val
            self.global_step = 0
            self.current_epoch = 0
            self.callback_metrics = {}

        def fit(self, module, train_dataloaders=None, datamodule=None):
            if datamodule is not None:
                datamodule.trainer = self        # the datamodule reads current_epoch / global_step (curriculum)
            loader = train_dataloaders if train_dataloaders is not None else datamodule.train_dataloader()
            optimizers, schedulers = module.configure_optimizers()
            opt = optimizers[0]
            module.__dict__["_shim_optimizers"] = list(optimizers)
            module.__dict__["_shim_schedulers"] = [s["scheduler"] for s in schedulers]
            module.train()
            manual = not getattr(module, "automatic_optimization", True)
            for _ in range(self.max_epochs):
                for bidx, batch in enumerate(loader):
                    if 0 <= self.max_steps <= self.global_step:
                        return
                    if manual:      # the module owns backward + optimizer step (pl manual optimization)
                        module.training_step(batch, bidx)
                    else:           # pl's closure order: training_step -> zero_grad -> backward -> step
                        out = module.training_step(batch, bidx)
                        opt.zero_grad(set_to_none=True)
                        out["loss"].backward()
                        if self.gradient_clip_val:
                            nn.utils.clip_grad_norm_(module.parameters(), self.gradient_clip_val)
                        opt.step()
                    self.global_step += 1
                    self.callback_metrics.update(getattr(module, "logged", {}))
                if manual:
                    module.on_train_epoch_end()
                else:
                    for s in schedulers:
                        s["scheduler"].step()
                self.current_epoch += 1
                if datamodule is not None and train_dataloaders is None:
                    loader = datamodule.train_dataloader()   # reload_dataloaders_every_n_epochs=1 (mbrl.py:361)

    pl = types.SimpleNamespace(LightningModule=LightningModule, LightningDataModule=LightningDataModule,
                               Callback=Callback, Trainer=Trainer,
                               callbacks=types.SimpleNamespace(Callback=Callback))
    IS_SHIM = True

__all__ = ["pl", "IS_SHIM"]
