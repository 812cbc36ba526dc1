from .bitcodec import encode_bitmap, decode_bitmap  # noqa: F401
