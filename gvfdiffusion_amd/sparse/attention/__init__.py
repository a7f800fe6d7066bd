from .full_attn import *        # noqa: F401,F403
from .serialized_attn import *  # noqa: F401,F403
from .windowed_attn import *    # noqa: F401,F403
from .modules import *          # noqa: F401,F403
