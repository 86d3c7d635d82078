"""Same logging convention as the reference (nellie/utils/base_logger.py:7-13): the module-level
`logger` is the `logging` module itself, configured at INFO with millisecond timestamps."""
import logging

logging.basicConfig(
    level=logging.INFO,
    format="%(asctime)s.%(msecs)03d :: %(levelname)s:%(name)s:[%(filename)s:%(lineno)d] :: %(message)s",
    datefmt="%Y-%m-%d | %H:%M:%S",
)
logger = logging
