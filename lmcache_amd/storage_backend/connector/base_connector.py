"""RemoteConnector -- the five-method byte-store interface of the reference
(lmcache/storage_backend/connector/base_connector.py:11-70)."""
import abc
from typing import List, Optional


class RemoteConnector(abc.ABC):
    @abc.abstractmethod
    def exists(self, key: str) -> bool:
        raise NotImplementedError

    @abc.abstractmethod
    def get(self, key: str) -> Optional[bytes]:
        raise NotImplementedError

    @abc.abstractmethod
    def set(self, key: str, obj: bytes) -> None:
        raise NotImplementedError

    @abc.abstractmethod
    def list(self) -> List[str]:
        raise NotImplementedError

    @abc.abstractmethod
    def close(self) -> None:
        raise NotImplementedError
