"""Operator-level Python bindings; importing this package registers every C-ABI signature."""
from .. import _lib  # noqa: F401
from . import dit_ops  # noqa: F401
from . import vae_ops  # noqa: F401
from ..sparse import vox2seq as _vox2seq  # noqa: F401,E402
