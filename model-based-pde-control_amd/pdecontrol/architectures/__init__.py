"""Surrogate-factory registry: ``--factory <Name>`` resolves ``getattr(pdecontrol.architectures, Name)``
(reference: pdecontrol/architectures/__init__.py:1-3, pdecontrol/mbrl/script.py:91)."""
from pdecontrol.architectures.autoreg import (KSAutoRegConvolutionalLSTM, KSAutoRegConvolutionalLSTMN,
                                              KSAutoRegFullyConnectedLSTM)
from pdecontrol.architectures.fno import BurgersFNO
