"""Shadow of the reference's `models` package that overrides only `models.losses`: every other submodule
(`models.evaluators`, `models.quadruplet_sentence_transformer`) is still found in the reference's own
`models/` directory further down sys.path."""
import pkgutil as _pkgutil

__path__ = _pkgutil.extend_path(__path__, __name__)
