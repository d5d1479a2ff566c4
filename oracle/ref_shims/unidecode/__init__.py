unidecode = lambda x: x
