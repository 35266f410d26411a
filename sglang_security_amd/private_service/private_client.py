"""SafeKV privacy hook: mirror of PrivateJudgeClient
(python/sglang/srt/managers/private_service/private_client.py:20-144).

The radix tree hands every node it creates to `update_privacy(node=, context=, prompt=)`.  A node whose parent is
already decided private inherits that decision at once (:73-77); otherwise the second-to-last line of the prompt
(:79-84) is queued for the detectors and a result thread later flips `node.private` / `node.need_check_privacy`
(:125-140).  The detectors themselves (the fork's rule / PII-model / LLM tiers) are out of scope: the judge is a
callable `detector(text) -> bool` (True = private) run on worker threads.  Node handles are the RadixCache
TreeNode handles, whose setters take the tree's lock in native code, so the result thread may mutate them while
the scheduler thread walks the tree.
"""
from __future__ import annotations

import queue
import threading
import time
from dataclasses import dataclass
from typing import Callable, Optional

_PLACEHOLDER = "hello, you are an simple assistant"  # private_client.py:85 (sic)


@dataclass
class PrivateNodeTask:
    """private_client.py task record (private_common.py)."""
    node: object
    task_type: str
    context: str
    prompt: str
    request_id: str
    timestamp: float
    privacy: bool


class PrivateJudgeClient:
    def __init__(self, server_args=None, port_args=None, detector: Optional[Callable[[str], bool]] = None,
                 num_workers: int = 1, batch_timeout: float = 0.05):
        """server_args / port_args: the reference's constructor arguments (private_client.py:32-35), kept for callers
        written against it; the detector service they configure there is control plane and is supplied here as
        `detector(text) -> True when the text is private`.  None: everything stays private (the fork's default when no
        detector answers: new nodes are private with need_check_privacy set, tree_node.py:50-52)."""
        self.server_args, self.port_args = server_args, port_args
        self.detector = detector
        self.batch_timeout = batch_timeout
        self.running = True
        self.task_queue: "queue.Queue[PrivateNodeTask]" = queue.Queue()
        self.result_queue: "queue.Queue[PrivateNodeTask]" = queue.Queue()
        self._threads = []
        if detector is not None:
            for _ in range(num_workers):
                t = threading.Thread(target=self._detect_task, daemon=True)
                t.start()
                self._threads.append(t)
        t = threading.Thread(target=self._response_task, daemon=True)
        t.start()
        self._threads.append(t)

    @staticmethod
    def judged_text(prompt: str) -> str:
        """The line the detectors see (private_client.py:79-85)."""
        lines = prompt.strip().split("\n")
        text = lines[-2] if len(lines) > 1 else lines[-1]
        if "Assistant" in text or "im_end" in text:
            text = _PLACEHOLDER
        return text

    def update_privacy(self, node, context: str, prompt: str) -> None:
        parent = node.parent
        if parent is not None and not parent.need_check_privacy and parent.private:
            node.need_check_privacy = False
            return
        if self.detector is None:
            return  # nobody consumes the queue: the node simply stays private and unchecked
        text = self.judged_text(prompt)
        self.task_queue.put(PrivateNodeTask(node=node, task_type="update_private", context=text, prompt=text,
                                            request_id="", timestamp=time.time(), privacy=True))

    def _detect_task(self) -> None:
        while self.running:
            try:
                task = self.task_queue.get(timeout=self.batch_timeout)
            except queue.Empty:
                continue
            try:
                task.privacy = bool(self.detector(task.prompt))
            except Exception:
                task.privacy = True  # a failing detector must not make anything public
            self.result_queue.put(task)

    def _response_task(self) -> None:
        while self.running:
            try:
                task = self.result_queue.get(timeout=self.batch_timeout)
            except queue.Empty:
                continue
            if task.prompt == _PLACEHOLDER:
                task.privacy = False
            try:
                if not task.privacy:
                    task.node.private = False
                task.node.need_check_privacy = False
            except KeyError:
                pass  # the node was evicted while the judgement was pending

    def drain(self, timeout: float = 5.0) -> bool:
        """Wait until every queued judgement has been applied (tests)."""
        end = time.time() + timeout
        while time.time() < end:
            if self.task_queue.empty() and self.result_queue.empty():
                time.sleep(2 * self.batch_timeout)
                if self.task_queue.empty() and self.result_queue.empty():
                    return True
            time.sleep(0.01)
        return False

    def close(self) -> None:
        self.running = False
