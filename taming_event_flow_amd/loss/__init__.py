"""Mirror of the reference's ``loss`` package.  When this package's parent directory is put on ``sys.path`` in front of the
reference checkout, modules that exist here shadow the reference's; everything else in the
reference's ``loss`` directory stays importable through the extended ``__path__``."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
