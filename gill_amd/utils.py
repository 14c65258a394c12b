"""The one helper of gill/utils.py that the generation path touches (gill/models.py:658, :759)."""


def truncate_caption(caption: str) -> str:
  """Truncate captions at periods and newlines.  (reference: gill/utils.py:32-40)"""
  caption = caption.strip('\n')
  trunc_index = caption.find('\n') + 1
  if trunc_index <= 0:
    trunc_index = caption.find('.') + 1
  if trunc_index > 0:
    caption = caption[:trunc_index]
  return caption
