"""Prompt builders with the reference's behaviour (merv/models/backbones/llm/prompting/base_prompter.py:28-73,
llama2_chat_prompter.py:30-88, vicuna_v15_prompter.py:22-79): the strings a checkpoint was trained on are a fixed contract,
pinned by tests/golden/prompts*.json (produced by the reference's own classes); the code around them is one small base."""
from __future__ import annotations

from typing import Optional

SYS_PROMPTS = {
    "merv": ("You are a helpful language and vision assistant. You are able to understand the visual content that the user "
             "provides, and assist the user with a variety of tasks using natural language."),
}


class PromptBuilder:
    """Turn bookkeeping shared by every builder: human / gpt turns alternate, `<image>` tags are dropped, the text handed to
    the tokenizer has no leading BOS (the tokenizer adds it) and no trailing space. Subclasses only say how a human turn, the
    FIRST human turn (where a system prompt goes, if the family has one) and an assistant turn are wrapped."""

    bos, eos = "<s>", "</s>"

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        self.model_family = model_family
        self.system_prompt = system_prompt
        self.prompt, self.turn_count = "", 0

    def _human(self, message: str) -> str:
        raise NotImplementedError

    def _first(self, message: str) -> str:
        return self._human(message)

    def _gpt(self, message: str) -> str:
        return f"{message if message != '' else ' '}{self.eos}"

    def _wrap_next(self, message: str) -> str:
        if self.turn_count % 2:
            return self._gpt(message)
        return self._first(message) if self.turn_count == 0 else self._human(message)

    def add_turn(self, role: str, message: str) -> str:
        expected = "gpt" if self.turn_count % 2 else "human"
        assert role == expected, f"turn {self.turn_count} must come from `{expected}`, got `{role}`"
        wrapped = self._wrap_next(message.replace("<image>", "").strip())
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, message: str) -> str:
        tail = self._first(message) if self.turn_count == 0 else self._human(message)
        return (self.prompt + tail).removeprefix(self.bos).rstrip()

    def get_prompt(self) -> str:
        return self.prompt.removeprefix(self.bos).rstrip()


class PurePromptBuilder(PromptBuilder):
    """base_prompter.py:28-73: `In: {msg}\nOut: ` / `{reply}</s>`, no system prompt."""

    def _human(self, message: str) -> str:
        return f"In: {message}\nOut: "


class LLaMa2ChatPromptBuilder(PromptBuilder):
    """llama2_chat_prompter.py:30-88: `<s>[INST] {msg} [/INST] `, the `<<SYS>` block inside the first instruction."""

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        text = SYS_PROMPTS[model_family] if system_prompt is None else system_prompt
        self.system_prompt = f"<<SYS>\n{text.strip()}\n<</SYS>>\n\n"  # (sic) the reference's opening tag

    def _human(self, message: str) -> str:
        return f"{self.bos}[INST] {message} [/INST] "

    def _first(self, message: str) -> str:
        return self._human(self.system_prompt + message)


VICUNA_SYS_PROMPTS = {
    "merv": ("A chat between a curious user and an artificial intelligence assistant. "
             "The assistant gives helpful, detailed, and polite answers to the user's questions."),
}


class VicunaV15ChatPromptBuilder(PromptBuilder):
    """vicuna_v15_prompter.py:22-79: `<system> USER: {msg} ASSISTANT: `, the system text once, before the first turn."""

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.system_prompt = (VICUNA_SYS_PROMPTS[self.model_family] if system_prompt is None else system_prompt).strip() + " "

    def _human(self, message: str) -> str:
        return f"USER: {message} ASSISTANT: "

    def _first(self, message: str) -> str:
        return self.system_prompt + self._human(message)


class MistralInstructPromptBuilder(PromptBuilder):
    """Mistral-7B-Instruct turns for the BASELINE.json configs[4] LLM swap: `[INST] {msg} [/INST] ` (the model card's
    template; no system-prompt slot, so a given system prompt is prepended to the first instruction)."""

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        super().__init__(model_family, system_prompt)
        self.system_prompt = "" if system_prompt is None else system_prompt.strip() + "\n\n"

    def _human(self, message: str) -> str:
        return f"[INST] {message} [/INST] "

    def _first(self, message: str) -> str:
        return self._human(self.system_prompt + message)


class _HeaderChatPromptBuilder(PromptBuilder):
    """The chat-template builders (llama2_chat_prompter.py:91-123, qwen2_prompter.py:11-42): the system block opens the prompt,
    a human turn is wrapped together with the assistant header that follows it, and -- unlike the older builders -- messages
    are taken verbatim (no `<image>` removal, no stripping) and get_prompt() returns the text as is. No system-prompt argument."""

    SYSTEM = HUMAN = GPT_END = ""

    def __init__(self, model_family: str, system_prompt: Optional[str] = None) -> None:
        if system_prompt is not None:
            raise TypeError(f"{type(self).__name__} takes no system_prompt (the reference's class has no such argument)")
        super().__init__(model_family, SYS_PROMPTS[model_family])
        self.prompt = self.SYSTEM.format(self.system_prompt)

    def add_turn(self, role: str, message: str) -> str:
        expected = "gpt" if self.turn_count % 2 else "human"
        assert role == expected, f"turn {self.turn_count} must come from `{expected}`, got `{role}`"
        wrapped = self.HUMAN.format(message) if role == "human" else message + self.GPT_END
        self.prompt += wrapped
        self.turn_count += 1
        return wrapped

    def get_potential_prompt(self, message: str):
        return None  # the reference's method only evaluates an assert on an exception instance and returns None

    def get_prompt(self) -> str:
        return self.prompt


class LLaMa31PromptBuilder(_HeaderChatPromptBuilder):
    SYSTEM = "<|start_header_id|>system<|end_header_id|>\n\n{}<|eot_id|>"  # <|begin_of_text|> comes from the tokenizer
    HUMAN = "<|start_header_id|>user<|end_header_id|>\n\n{}<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"
    GPT_END = "<|eot_id|>"


class Qwen2PromptBuilder(_HeaderChatPromptBuilder):
    SYSTEM = "<|im_start|>system\n{}<|im_end|>\n"
    HUMAN = "<|im_start|>user\n{}<|im_end|>\n<|im_start|>assistant\n"
    GPT_END = "<|im_end|>"
