from .backend import ForwardMetadata, SglkAttnBackend  # noqa: F401
