"""Conversation template with the behaviour of the reference's ``data/conversation/lib.py`` (:25-121):
``system\\n\\n### Human: q\\n### Assistant: a\\n###`` (SINGLE) and the two-separator style.  Host-side only."""
from __future__ import annotations

import dataclasses
from enum import Enum, auto
from typing import List, Optional, Tuple


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()


@dataclasses.dataclass
class Conversation:
    system: str
    roles: Tuple[str, str]
    messages: List
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    version: str = "Unknown"
    skip_next: bool = False

    def process(self):
        """{"conv": full text, "to_predict": assistant spans the model learns} (lib.py:25-56)."""
        to_predict = []
        last = len(self.messages) - 1
        if self.sep_style == SeparatorStyle.SINGLE:
            ret = self.system + "\n\n" + self.sep
            for i, (role, message) in enumerate(self.messages):
                if message is not None:
                    if type(message) is tuple:
                        message = message[0]
                    ret += " " + role + ": " + message + "\n" + self.sep
                    if role == self.roles[1]:
                        to_predict.append(message + "\n" + self.sep)
                else:
                    assert i == last, "only last message can be None"
                    ret += " " + role + ":"
        elif self.sep_style == SeparatorStyle.TWO:
            seps = [self.sep, self.sep2]
            ret = self.system + seps[0]
            for i, (role, message) in enumerate(self.messages):
                if message:
                    if type(message) is tuple:
                        message = message[0]
                    ret += " " + role + ": " + message + seps[i % 2]
                    if role == self.roles[1]:
                        to_predict.append(message + seps[i % 2])
                else:
                    assert i == last, "only last message can be None"
                    ret += " " + role + ":"
        else:
            raise ValueError(f"Invalid style: {self.sep_style}")
        return {"conv": ret, "to_predict": to_predict}

    def get_prompt(self) -> str:
        return self.process()["conv"]

    def append_message(self, role, message) -> None:
        self.messages.append([role, message])

    def copy(self) -> "Conversation":
        return Conversation(system=self.system, roles=self.roles, messages=[[x, y] for x, y in self.messages],
                            sep_style=self.sep_style, sep=self.sep, sep2=self.sep2)

    def load_qas(self, qas) -> None:
        self.messages = []
        for q, a in qas:
            self.append_message(self.roles[0], q)
            self.append_message(self.roles[1], a)

    @property
    def response_end_signal(self) -> str:
        return "\n" + self.sep if self.sep_style == SeparatorStyle.SINGLE else self.sep2


def conv_v1_2() -> Conversation:
    return Conversation(
        system="A chat between a curious human and an artificial intelligence assistant. "
               "The assistant gives helpful, detailed, and polite answers to the human's questions.",
        roles=("Human", "Assistant"), messages=[], sep_style=SeparatorStyle.SINGLE, sep="###")


default_conversation = conv_v1_2
