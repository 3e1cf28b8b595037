from .lib import Conversation, SeparatorStyle, conv_v1_2, default_conversation  # noqa: F401
