"""Base class of the surrogate factories selected by ``--factory`` (mirror of the reference's
``pdecontrol/surrogates/factory.py`` :11-34)."""
from pdecontrol._compat.munch import munchify
from pdecontrol.surrogates.surrogate import PDESurrogate


class PDESurrogateFactory:
    def surrogate(self, **kwargs):
        return PDESurrogate(**kwargs)

    def model(self, **kwargs):
        raise NotImplementedError

    @property
    def defaults(self):
        """Per-section default configs, overridden by the CLI's JSON flags (script.py:94-109)."""
        return munchify({"model": {}, "surrogate": {}, "training": {}, "trainer": {}, "curriculum": {}})
