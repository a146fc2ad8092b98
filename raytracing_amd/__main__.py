from .renderer import main

main()
