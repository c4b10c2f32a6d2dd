"""Prompt builders with the reference's behaviour (merv/models/backbones/llm/prompting/base_prompter.py:28-73,
llama2_chat_prompter.py:30-88): same wrap strings, same turn bookkeeping, same get_prompt() stripping."""
from __future__ import annotations

from typing import Optional

SYS_PROMPTS = {
    "merv": ("You are a helpful language and vision assistant. You are able to understand the visual content that the user "
             "provides, and assist the user with a variety of tasks using natural language."),
}


class PromptBuilder:
    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        self.model_family = model_family
        self.system_prompt = system_prompt


class PurePromptBuilder(PromptBuilder):
    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.bos, self.eos = "<s>", "</s>"
        self.wrap_human = lambda msg: f"In: {msg}\nOut: "
        self.wrap_gpt = lambda msg: f"{msg if msg != '' else ' '}{self.eos}"
        self.prompt, self.turn_count = "", 0

    def add_turn(self, role: str, message: str) -> str:
        assert (role == "human") if (self.turn_count % 2 == 0) else (role == "gpt")
        message = message.replace("<image>", "").strip()
        wrapped = self.wrap_human(message) if (self.turn_count % 2) == 0 else self.wrap_gpt(message)
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, message: str) -> str:
        return (str(self.prompt) + self.wrap_human(message)).removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        return self.prompt.removeprefix(self.bos).rstrip()


def format_system_prompt(system_prompt: str) -> str:
    return f"<<SYS>\n{system_prompt.strip()}\n<</SYS>>\n\n"  # (sic) the reference's opening tag


class LLaMa2ChatPromptBuilder(PromptBuilder):
    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.system_prompt = format_system_prompt(SYS_PROMPTS[self.model_family] if system_prompt is None else system_prompt)
        self.bos, self.eos = "<s>", "</s>"
        self.wrap_human = lambda msg: f"{self.bos}[INST] {msg} [/INST] "
        self.wrap_gpt = lambda msg: f"{msg if msg != '' else ' '}{self.eos}"
        self.prompt, self.turn_count = "", 0

    def add_turn(self, role: str, message: str) -> str:
        assert (role == "human") if (self.turn_count % 2 == 0) else (role == "gpt")
        message = message.replace("<image>", "").strip()
        if self.turn_count == 0:
            wrapped = self.wrap_human(self.system_prompt + message)
        elif (self.turn_count % 2) == 0:
            wrapped = self.wrap_human(message)
        else:
            wrapped = self.wrap_gpt(message)
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, user_msg: str) -> str:
        prompt_copy = str(self.prompt)
        prompt_copy += self.wrap_human((self.system_prompt + user_msg) if self.turn_count == 0 else user_msg)
        return prompt_copy.removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        return self.prompt.removeprefix(self.bos).rstrip()


class _SystemFirstTurnBuilder(PromptBuilder):
    """Chat builders whose system prompt is prepended to the first human turn only."""

    bos, eos = "<s>", "</s>"

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.prompt, self.turn_count = "", 0

    def _first(self, message: str) -> str:
        raise NotImplementedError

    def _human(self, message: str) -> str:
        raise NotImplementedError

    def _gpt(self, message: str) -> str:
        return f"{message if message != '' else ' '}{self.eos}"

    def add_turn(self, role: str, message: str) -> str:
        assert (role == "human") if (self.turn_count % 2 == 0) else (role == "gpt")
        message = message.replace("<image>", "").strip()
        if self.turn_count == 0:
            wrapped = self._first(message)
        else:
            wrapped = self._human(message) if self.turn_count % 2 == 0 else self._gpt(message)
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, message: str) -> str:
        tail = self._first(message) if self.turn_count == 0 else self._human(message)
        return (str(self.prompt) + tail).removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        return self.prompt.removeprefix(self.bos).rstrip()


VICUNA_SYS_PROMPTS = {
    "merv": ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions."),
}


class VicunaV15ChatPromptBuilder(_SystemFirstTurnBuilder):
    """vicuna_v15_prompter.py:22-79: `<system> USER: {msg} ASSISTANT: `, the system text once, before the first turn."""

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.system_prompt = (VICUNA_SYS_PROMPTS[self.model_family] if system_prompt is None else system_prompt).strip() + " "

    def _human(self, message: str) -> str:
        return f"USER: {message} ASSISTANT: "

    def _first(self, message: str) -> str:
        return self.system_prompt + self._human(message)


class MistralInstructPromptBuilder(_SystemFirstTurnBuilder):
    """Mistral-7B-Instruct turns for the BASELINE.json configs[4] LLM swap: `[INST] {msg} [/INST] ` (the model card's
    template; no system-prompt slot, so a given system prompt is prepended to the first instruction)."""

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.system_prompt = "" if system_prompt is None else system_prompt.strip() + "\n\n"

    def _human(self, message: str) -> str:
        return f"[INST] {message} [/INST] "

    def _first(self, message: str) -> str:
        return self._human(self.system_prompt + message)
